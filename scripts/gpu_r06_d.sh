#!/bin/bash
# round 6, call D: scheduler tests (strided plan, timeout report), pre-pack cache tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-d}
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -k "scheduler or prepack or strided or hand_over or plan" > $O/pytest_d_$T.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_d_$T.log | cut -c1-400
