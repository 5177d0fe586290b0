#!/bin/bash
# round 4: general strides test, scheduler file, full suite, every BASELINE config line, default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-i}
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 600 python -m pytest tests/test_gpu_parity.py -q --timeout 300 -k "any_matrix_view or transposed_b_4096" > $O/strides_tests_$T.log 2>&1; echo "strides rc=$?"; grep -v "$F" $O/strides_tests_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | tail -12
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -30
timeout 600 python scripts/bench_configs.py > $O/configs_$T.jsonl 2> $O/configs_$T.err; echo "configs rc=$?"; cut -c1-230 $O/configs_$T.jsonl
timeout 400 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; echo "bench rc=$?"; tail -1 $O/bench_$T.json | cut -c1-3000; tail -3 $O/bench_$T.err
