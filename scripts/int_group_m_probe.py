#!/usr/bin/env python3
"""Raster group of the hand-scheduled integer limb kernels (option int_group_m) through the API, packing pass included: int32 / int64 n^3,
every value timed in interleaved rounds, results compared with the first value's.   usage: int_group_m_probe.py [values ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from scripts.bench_configs import ev_time
vals = [int(x) for x in sys.argv[1:]] or [8, 4, 2, 16]
for dt, n in [(torch.int32, 8192), (torch.int32, 4096), (torch.int32, 2048), (torch.int64, 8192), (torch.int64, 4096), (torch.int64, 1920)]:
    lim = 2**31 if dt == torch.int32 else 2**62
    A = torch.randint(-lim, lim - 1, (n, n), device="cuda", dtype=dt)
    B = torch.randint(-lim, lim - 1, (n, n), device="cuda", dtype=dt)
    C = torch.zeros((n, n), device="cuda", dtype=dt)
    rec = {"dtype": str(dt).split(".")[1], "n": n, "tintops": {}, "same": True}
    ref = None
    ms = {v: [] for v in vals}
    for r in range(3):
        for v in vals:
            laser_amd.set_option("int_group_m", v)
            t, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=5)
            ms[v].append(t)
            if ref is None:
                ref = C.clone()
            rec["same"] = rec["same"] and bool(torch.equal(C, ref))
    for v in vals:
        rec["tintops"][str(v)] = round(2.0 * n ** 3 / sorted(ms[v])[1] / 1e9, 1)
    print(json.dumps(rec), flush=True)
laser_amd.set_option("int_group_m", 4)
