#!/bin/bash
# round 5, call B: the whole GPU suite again (no -x), im2col with the 16-byte LDS reads, the small-channel conv A/B (8- vs 16-byte
# stores), rocprofv3 evidence for C3 and C4 on the current sources (warm-only averages in the summaries)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-b}
O=gpurun_out/r05; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | tail -30
timeout 120 python scripts/im2col_probe.py > $O/im2col_probe_$T.jsonl 2> $O/im2col_probe_$T.err; cat $O/im2col_probe_$T.jsonl | cut -c1-260; tail -3 $O/im2col_probe_$T.err
timeout 120 python scripts/conv_small_ab.py > $O/conv_small_ab_$T.jsonl 2> $O/conv_small_ab_$T.err; cat $O/conv_small_ab_$T.jsonl | cut -c1-400; tail -3 $O/conv_small_ab_$T.err
timeout 600 bash scripts/gpu_profile_cmd.sh c3 python scripts/c3_run.py 40 > /dev/null 2>&1; rm -rf $O/rocprof_c3; cp -r gpurun_out/prof_c3 $O/rocprof_c3; head -12 $O/rocprof_c3/summary.md; tail -3 $O/rocprof_c3/stats.log
timeout 600 bash scripts/gpu_profile_cmd.sh c4 python scripts/conv_c4_run.py 10 > /dev/null 2>&1; rm -rf $O/rocprof_c4; cp -r gpurun_out/prof_c4 $O/rocprof_c4; head -16 $O/rocprof_c4/summary.md; tail -3 $O/rocprof_c4/stats.log
