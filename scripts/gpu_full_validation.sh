#!/bin/bash
# full validation of the tree on one MI355X (gpurun) -- whole GPU suite, smoke, bench line, headline profile (+ traffic file on
# the same tree), C4 conv profile (kernel-trace + PMC), every BASELINE config line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_full.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_full.log | tail -12
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -1 $O/bench_full.json | cut -c1-1500; tail -2 $O/bench_full.err
timeout 900 bash scripts/gpu_profile_bench.sh default > /dev/null 2>&1; rm -rf $O/rocprof_bench_default; cp -r gpurun_out/prof_default $O/rocprof_bench_default; head -8 $O/rocprof_bench_default/summary.md; head -5 $O/rocprof_bench_default/pmc_traffic.json | cut -c1-200
timeout 900 bash scripts/gpu_profile_cmd.sh conv_c4 python scripts/conv_c4_run.py 4 > /dev/null 2>&1; rm -rf $O/rocprof_conv_c4; cp -r gpurun_out/prof_conv_c4 $O/rocprof_conv_c4; head -16 $O/rocprof_conv_c4/summary.md
timeout 900 python scripts/bench_configs.py > $O/configs_full.jsonl 2> $O/configs_full.err; cat $O/configs_full.jsonl; tail -3 $O/configs_full.err
