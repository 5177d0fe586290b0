#!/bin/bash
# round 4: the scheduler tests (three times: the concurrency cases are timing-dependent), the launch-plan sweeps, the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-g}
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_scheduler.py -q --timeout 300 > $O/sched_tests_${T}_$i.log 2>&1; echo "sched[$i] rc=$?"; grep -v "$F" $O/sched_tests_${T}_$i.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -8; done
timeout 200 python scripts/concurrency_probe.py > $O/concurrency_probe_$T.jsonl 2> $O/concurrency_probe_$T.err; echo "probe rc=$?"; grep -c '"wrong": 0, "first_wrong": \[\], "timeouts_total": 0' $O/concurrency_probe_$T.jsonl
timeout 300 python scripts/plan_sweep.py f32 mid > $O/plan_sweep_f32_mid_$T.jsonl 2> $O/plan_sweep_$T.err; echo "sweep rc=$?"
timeout 200 python scripts/plan_sweep.py f64 f64 > $O/plan_sweep_f64_$T.jsonl 2>> $O/plan_sweep_$T.err; echo "sweep64 rc=$?"
timeout 120 python scripts/plan_sweep.py f32 big > $O/plan_sweep_f32_big_$T.jsonl 2>> $O/plan_sweep_$T.err; echo "sweepbig rc=$?"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -40
