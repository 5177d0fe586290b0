#!/bin/bash
# round 2, GPU call 3: small-matrix path (parity + compiled-caller timing), concurrent tail, full suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_v3.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v3.log; tail -30 $O/pytest_gpu_v3.log
L=laser_amd/lib; g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/small_gemm_bench.cpp -o /tmp/small_gemm_bench -L$L -llaser_hip -Wl,-rpath,$PWD/$L -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 && (timeout 120 /tmp/small_gemm_bench; timeout 120 /tmp/small_gemm_bench) > $O/small_gemm_v1.jsonl 2>&1; cat $O/small_gemm_v1.jsonl
timeout 300 python scripts/small_path_probe.py > $O/small_path_probe_v1.log 2>&1; cat $O/small_path_probe_v1.log
timeout 300 python scripts/conv_c4_run.py 10 > $O/conv_c4_v3.log 2>&1; cat $O/conv_c4_v3.log
timeout 300 python scripts/heuristic_check.py 256x100352x1152,5000x5000x5000,4100x4100x4100 > $O/heuristic_check_v3.jsonl 2> $O/heuristic_check_v3.err; python - <<'PY'
import json
for l in open('gpurun_out/r02/heuristic_check_v3.jsonl'):
    d=json.loads(l); print(d['shape'], d['mode'][:5], d['chosen'], 'cut',d['cut'], 'auto',d['auto_ms'],'nosplit',d['auto_nosplit_ms'],'best',d['best'],d['best_ms'],'TF',d['auto_tflops'],'ratio',d['auto_over_best'])
PY
