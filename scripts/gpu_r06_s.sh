#!/bin/bash
# round 6, call S: column stride on C through the 16x16-block tiles; parity + scheduler + tensor files; fuzz
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-s}
O=gpurun_out/r06; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed" | tail -12
timeout 900 python scripts/fuzz_gemm.py 600 83 > $O/fuzz_gemm_$T.log 2>&1; echo "fuzz gemm rc=$?"; tail -2 $O/fuzz_gemm_$T.log | cut -c1-300
