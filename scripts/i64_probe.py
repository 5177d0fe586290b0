#!/usr/bin/env python3
"""int64 GEMM: eight int8 limbs on the matrix cores (gemm_i64_mfma.hip) vs the VALU kernel, full-range operands.
One JSON line per (size, path); equality of the two paths is asserted on every size."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from scripts.bench_configs import ev_time

for n in (256, 512, 960, 1920, 4096, 8192):
    A = torch.randint(-2 ** 62, 2 ** 62, (n, n), device="cuda", dtype=torch.int64)
    B = torch.randint(-2 ** 62, 2 ** 62, (n, n), device="cuda", dtype=torch.int64)
    C = {True: torch.zeros((n, n), device="cuda", dtype=torch.int64), False: torch.zeros((n, n), device="cuda", dtype=torch.int64)}
    for on in (True, False):
        if not on and n > 4096:
            continue
        laser_amd.set_i64_mfma(on)
        med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C[on]), iters=5 if n > 2048 else 9)
        print(json.dumps({"config": f"gemm int64 {n}^3 " + ("(int8-limb MFMA, 36 limb products)" if on else "(VALU kernel)"),
                          "ms_med": round(med, 4), "ms_min": round(mn, 4), "tops": round(2.0 * n ** 3 / (med * 1e-3) / 1e12, 3)}), flush=True)
    laser_amd.set_i64_mfma(True)
    if n <= 4096:
        assert torch.equal(C[True], C[False]), n
