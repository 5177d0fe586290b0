#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu8.log
timeout 900 python scripts/bench_configs.py > gpurun_out/configs.log 2>&1; echo "configs rc=$?"; grep '^{' gpurun_out/configs.log | grep -i "host-pointer\|C2\|C4 conv"
