#!/bin/bash
# round 2, GPU call 13: completion-flag polling on the small zero-copy host path (parity + timing A/B), pinned-only 2-D pipeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "host_pointer or small or smoke or cpp or semantics or oracle" > $O/pytest_gpu_v13_small.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v13_small.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v13_small.log | tail -12
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
L=laser_amd/lib; g++ -std=c++17 -O2 -w -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/small_gemm_bench.cpp -o /tmp/small_gemm_bench -L$L -llaser_hip -Wl,-rpath,$PWD/$L -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 && (timeout 120 /tmp/small_gemm_bench; timeout 120 /tmp/small_gemm_bench) > $O/small_gemm_v13.jsonl 2>&1; cat $O/small_gemm_v13.jsonl
timeout 300 python scripts/host_pipeline_probe.py > $O/host_pipeline_v3.jsonl 2>&1; cat $O/host_pipeline_v3.jsonl
