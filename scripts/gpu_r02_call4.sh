#!/bin/bash
# round 2, GPU call 4: full parity suite (small-matrix path, forEach twin, conv workspace contract, fuzz with the new
# shape classes), compiled-caller timing of C1, every BASELINE config line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_v4.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v4.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v4.log | tail -40
L=laser_amd/lib; g++ -std=c++17 -O2 -w -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/small_gemm_bench.cpp -o /tmp/small_gemm_bench -L$L -llaser_hip -Wl,-rpath,$PWD/$L -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 && (timeout 120 /tmp/small_gemm_bench; timeout 120 /tmp/small_gemm_bench) > $O/small_gemm_v2.jsonl 2>&1; cat $O/small_gemm_v2.jsonl
timeout 300 python scripts/small_path_probe.py > $O/small_path_probe_v2.log 2>&1; cat $O/small_path_probe_v2.log
timeout 900 python scripts/bench_configs.py > $O/configs_v1.jsonl 2> $O/configs_v1.err; cat $O/configs_v1.jsonl; tail -3 $O/configs_v1.err
