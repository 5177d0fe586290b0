#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu19.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu19.log | cut -c1-300
timeout 600 python scripts/conv_cfg_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_probe.log
