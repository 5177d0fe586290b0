#!/bin/bash
# round 2, GPU call 5 (re-entry): full parity suite at HEAD, bench line, rocprofv3 kernel-trace + PMC passes of the bench
# command, compiled-caller small-matrix timing, every BASELINE config line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_v5.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v5.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v5.log | tail -15
timeout 400 python bench.py > $O/bench_v5.json 2> $O/bench_v5.err; tail -1 $O/bench_v5.json; tail -3 $O/bench_v5.err
timeout 900 bash scripts/gpu_profile_bench.sh default > /dev/null 2>&1; rm -rf $O/rocprof_bench_default; cp -r gpurun_out/prof_default $O/rocprof_bench_default; head -50 $O/rocprof_bench_default/summary.md
L=laser_amd/lib; g++ -std=c++17 -O2 -w -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/small_gemm_bench.cpp -o /tmp/small_gemm_bench -L$L -llaser_hip -Wl,-rpath,$PWD/$L -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 && (timeout 120 /tmp/small_gemm_bench; timeout 120 /tmp/small_gemm_bench) > $O/small_gemm_v5.jsonl 2>&1; cat $O/small_gemm_v5.jsonl
timeout 900 python scripts/bench_configs.py > $O/configs_v5.jsonl 2> $O/configs_v5.err; cat $O/configs_v5.jsonl; tail -3 $O/configs_v5.err
