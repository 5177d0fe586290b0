#!/usr/bin/env python3
"""Forced-configuration timing at 8192^3 (and 4096^3): usage cfg_probe.py cfg:mode [cfg:mode ...]  (mode 0 laser-order, 1 fast)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from scripts.bench_configs import ev_time
names = laser_amd.f32_configs()
for n in (8192, 4096):
    A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; C = torch.zeros((n, n), device="cuda")
    ref = {}
    for spec in sys.argv[1:]:
        cfg, mode = (int(v) for v in spec.split(":"))
        laser_amd.set_f32_config(cfg); laser_amd.set_float_mode(mode)
        med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7)
        same = None
        if mode == 0:
            if 0 not in ref: ref[0] = C.clone()
            same = bool(torch.equal(ref[0], C))
        print(json.dumps({"n": n, "cfg": names[cfg] if cfg >= 0 else "heuristic", "mode": "laser_order" if mode == 0 else "fast", "ms": round(med, 4),
                          "tflops": round(2.0 * n ** 3 / med / 1e9, 1), "bit_identical_to_first_laser": same}), flush=True)
laser_amd.set_f32_config(-1); laser_amd.set_float_mode(0)
