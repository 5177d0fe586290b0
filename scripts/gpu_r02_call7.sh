#!/bin/bash
# round 2, GPU call 7: conv K-slice-parallel tail (parity + C4 timing A/B), int64 limb kernel as a 3-stage DMA ring
# (parity + timing), headline profile with the traffic file generated on the same tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "conv or int64 or int32 or split_tail" > $O/pytest_gpu_v7_conv_int.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v7_conv_int.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v7_conv_int.log | tail -30
timeout 300 python scripts/i64_probe.py > $O/i64_probe_v2.jsonl 2>&1; cat $O/i64_probe_v2.jsonl
timeout 300 python scripts/conv_c4_run.py 10 > $O/conv_c4_v7.log 2>&1; cat $O/conv_c4_v7.log
timeout 900 bash scripts/gpu_profile_bench.sh default > /dev/null 2>&1; rm -rf $O/rocprof_bench_default; cp -r gpurun_out/prof_default $O/rocprof_bench_default; head -12 $O/rocprof_bench_default/summary.md; cat $O/rocprof_bench_default/pmc_traffic.json | head -12
