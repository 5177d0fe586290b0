#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu9.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu9.log
timeout 900 python scripts/bench_configs.py > gpurun_out/configs.log 2>&1; echo "configs rc=$?"; grep '^{' gpurun_out/configs.log | grep -i "float64\|C2 fp32 8192^3 cont"
