#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-j}
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py -q --timeout 600 -k "sharded or any_matrix_view or transposed_b_4096 or f32_asm_kernels_bit_exact" > $O/tests_$T.log 2>&1; echo "tests rc=$?"; grep -v "$F" $O/tests_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -12
timeout 300 python bench.py --single-process 1 --steps 10 --warmup 3 > $O/single_process_1gpu_$T.json 2> $O/single_process_1gpu_$T.err; echo "sp rc=$?"; tail -1 $O/single_process_1gpu_$T.json | cut -c1-900
timeout 100 python scripts/shape_run.py 8192 8192 8192 0 -1 0 20 | tail -1
timeout 600 bash scripts/group_m_sweep.sh 2>&1 | tail -8
