#!/usr/bin/env python3
"""Ragged-by-a-few shapes: peeled (default) vs single launch plan (set_split_tail(0)), both accumulation modes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from scripts.bench_configs import ev_time
for (M, N, K) in [(4100, 4100, 4100), (4095, 4097, 4099), (8200, 8200, 8192), (4100, 4096, 4096), (1000, 3000, 2000)]:
    A = (torch.rand((M, K), device="cuda") - 0.5) * 0.2; B = (torch.rand((K, N), device="cuda") - 0.5) * 0.2; C = torch.zeros((M, N), device="cuda")
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        r = {}
        for plan in (1, 0):
            laser_amd.set_split_tail(plan)
            med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7)
            r[plan] = med
        laser_amd.set_split_tail(1)
        print(json.dumps({"shape": [M, N, K], "mode": "laser_order" if mode == 0 else "fast", "ms": round(r[1], 4), "ms_single_plan": round(r[0], 4),
                          "tflops": round(2.0 * M * N * K / r[1] / 1e9, 1), "frac_mfma_peak": round(2.0 * M * N * K / r[1] / 1e9 / 157.3, 4)}), flush=True)
laser_amd.set_float_mode(0)
