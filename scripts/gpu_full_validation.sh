#!/bin/bash
# full validation of the tree on one MI355X (gpurun): whole GPU suite, smoke, bench line, headline profile (+ the traffic file made on
# the same tree), every BASELINE config line, the launch-plan sweeps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-final}
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed" | tail -12
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1000 bash scripts/gpu_profile_bench.sh default > /dev/null 2>&1; rm -rf $O/rocprof_bench_default; cp -r gpurun_out/prof_default $O/rocprof_bench_default; head -12 $O/rocprof_bench_default/summary.md; cat $O/rocprof_bench_default/pmc_traffic.json | cut -c1-300 | head -12
cp $O/rocprof_bench_default/pmc_traffic.json profiles/pmc_traffic.json
timeout 500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; tail -1 $O/bench_$T.json | cut -c1-1200; tail -2 $O/bench_$T.err
timeout 120 python scripts/prologue_cost.py > $O/prologue_cost_$T.jsonl 2> $O/prologue_cost_$T.err; cat $O/prologue_cost_$T.jsonl | cut -c1-300
timeout 200 python scripts/probes/conv_direct_probe.py > $O/conv_direct_probe_$T.jsonl 2> $O/conv_direct_probe_$T.err; cut -c1-200 $O/conv_direct_probe_$T.jsonl
timeout 900 python scripts/bench_configs.py > $O/configs_$T.jsonl 2> $O/configs_$T.err; wc -l $O/configs_$T.jsonl
timeout 300 python scripts/plan_sweep.py f32 mid > $O/plan_sweep_f32_mid_$T.jsonl 2> $O/plan_sweep_$T.err
timeout 200 python scripts/plan_sweep.py f64 f64 > $O/plan_sweep_f64_$T.jsonl 2>> $O/plan_sweep_$T.err
timeout 120 python scripts/plan_sweep.py f32 big > $O/plan_sweep_f32_big_$T.jsonl 2>> $O/plan_sweep_$T.err
