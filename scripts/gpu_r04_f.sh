#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-m}
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 900 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_parity.py -q --timeout 600 -k "scheduler or asm or full_size or matrix_view" > $O/tests_$T.log 2>&1; echo "tests rc=$?"; grep -v "$F" $O/tests_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -12
timeout 300 python scripts/plan_sweep.py f32 mid > $O/plan_sweep_f32_mid_$T.jsonl 2> $O/plan_sweep_$T.err; echo "sweep rc=$?"
timeout 200 python scripts/plan_sweep.py f64 f64 > $O/plan_sweep_f64_$T.jsonl 2>> $O/plan_sweep_$T.err; echo "sweep64 rc=$?"
timeout 200 python scripts/plan_sweep.py f32 small > $O/plan_sweep_f32_small_$T.jsonl 2>> $O/plan_sweep_$T.err; echo "sweepsmall rc=$?"
timeout 120 python scripts/plan_sweep.py f32 big > $O/plan_sweep_f32_big_$T.jsonl 2>> $O/plan_sweep_$T.err; echo "sweepbig rc=$?"
