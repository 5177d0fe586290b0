#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu10.log
python scripts/shape_probe.py 2>&1 | grep -v amdgpu
python scripts/conv_cfg_probe.py 2>&1 | grep -v amdgpu
