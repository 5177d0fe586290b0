#!/bin/bash
# round 4, call A: the new scheduler tests first (bounded), then the launch-plan sweep, then the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 600 python -m pytest tests/test_gpu_scheduler.py -q --timeout 300 > $O/sched_tests_f.log 2>&1; echo "sched rc=$?"; grep -v "$F" $O/sched_tests_f.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -30
timeout 300 python scripts/plan_sweep.py f32 mid > $O/plan_sweep_f32_mid_f.jsonl 2> $O/plan_sweep_f.err; echo "sweep rc=$?"; tail -2 $O/plan_sweep_f.err
timeout 200 python scripts/plan_sweep.py f64 f64 > $O/plan_sweep_f64_f.jsonl 2>> $O/plan_sweep_f.err; echo "sweep64 rc=$?"
timeout 120 python scripts/plan_sweep.py f32 big > $O/plan_sweep_f32_big_f.jsonl 2>> $O/plan_sweep_f.err; echo "sweepbig rc=$?"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_f.log 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu_f.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -40
