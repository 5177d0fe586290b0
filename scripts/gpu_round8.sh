#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tensor.py -q -x > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu8.log | cut -c1-250
