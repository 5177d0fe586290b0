#!/bin/bash
# round 2, GPU call 10: tiled limb-planes pass (parity of both integer paths + timing)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "int64 or int32 or fuzz or bit_exact_vs_oracle" > $O/pytest_gpu_v10_int.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v10_int.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v10_int.log | tail -30
timeout 300 python scripts/i64_probe.py > $O/i64_probe_v3.jsonl 2>&1; cat $O/i64_probe_v3.jsonl
timeout 300 python - > $O/i32_probe_v1.jsonl 2>&1 <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, laser_amd
from scripts.bench_configs import ev_time
for n in (960, 1920, 4096, 8192):
    A = torch.randint(-2**30, 2**30, (n, n), device="cuda", dtype=torch.int32); B = torch.randint(-2**30, 2**30, (n, n), device="cuda", dtype=torch.int32)
    C = torch.zeros((n, n), device="cuda", dtype=torch.int32)
    med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7)
    Bt = B.t().contiguous().t()   # column-major B: the k-contiguous source layout for the B planes
    med2, _ = ev_time(lambda: laser_amd.matmul(A, Bt, 1, 0, C), iters=7)
    print(json.dumps({"config": f"gemm int32 {n}^3 (int8-limb MFMA)", "ms_med": round(med, 4), "tops": round(2.0*n**3/(med*1e-3)/1e12, 2), "B_colmajor_ms": round(med2, 4)}), flush=True)
PY
cat $O/i32_probe_v1.jsonl
bash scripts/gpu_profile_cmd.sh int_limb python scripts/int_gemm_run.py 4 > /dev/null 2>&1; head -12 gpurun_out/prof_int_limb/summary.md; rm -rf $O/rocprof_int_limb; cp -r gpurun_out/prof_int_limb $O/rocprof_int_limb
