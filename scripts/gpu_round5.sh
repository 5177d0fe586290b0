#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "conv or im2col or kat" > gpurun_out/pytest_gpu5.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu5.log
timeout 600 python scripts/conv_cfg_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_probe.log
