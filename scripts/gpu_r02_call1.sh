#!/bin/bash
# round 2, GPU call 1: parity suite, bench line, main-loop ablation, C4 conv timing + rocprofv3 counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
lscpu > $O/host.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_v1.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v1.log; tail -3 $O/pytest_gpu_v1.log
timeout 300 python bench.py > $O/bench_v1.json 2> $O/bench_v1.err; tail -1 $O/bench_v1.json
timeout 300 python scripts/ablate_f32.py > $O/ablate_v1.log 2>&1; cat $O/ablate_v1.log
timeout 300 python scripts/conv_c4_run.py 10 > $O/conv_c4_v1.log 2>&1; cat $O/conv_c4_v1.log
timeout 300 python scripts/heuristic_check.py > $O/heuristic_check_v1.jsonl 2> $O/heuristic_check_v1.err; tail -40 $O/heuristic_check_v1.jsonl
timeout 900 bash scripts/gpu_profile_cmd.sh conv_c4 python scripts/conv_c4_run.py 4; cp -r gpurun_out/prof_conv_c4 $O/rocprof_conv_c4; head -60 $O/rocprof_conv_c4/summary.md
