#!/bin/bash
# round 2, GPU call 6: int64 on the int8 matrix cores (parity + timing), headline-only rocprofv3 profile (kernel-trace +
# PMC passes) with the traffic file regenerated from it, C4 conv loader A/B, float64 slice-parallel eligibility probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "int64 or int32 or bit_exact_vs_oracle or device_resident" > $O/pytest_gpu_v6_int.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v6_int.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v6_int.log | tail -30
timeout 300 python scripts/i64_probe.py > $O/i64_probe_v1.jsonl 2>&1; cat $O/i64_probe_v1.jsonl
timeout 900 bash scripts/gpu_profile_bench.sh default > /dev/null 2>&1; rm -rf $O/rocprof_bench_default; cp -r gpurun_out/prof_default $O/rocprof_bench_default; head -70 $O/rocprof_bench_default/summary.md
timeout 300 python scripts/conv_c4_run.py 10 > $O/conv_c4_v6.log 2>&1; cat $O/conv_c4_v6.log
timeout 300 python scripts/f64_slice_probe.py > $O/f64_slice_probe_v1.jsonl 2>&1; cat $O/f64_slice_probe_v1.jsonl
