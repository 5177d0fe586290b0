#!/usr/bin/env python3
"""Tuning sweep (GPU box): every f32 tile configuration x accumulation mode x operand layout on the
BASELINE shapes, timed with HIP events on the launch stream.  Interleaved rounds (guide rule 24)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd

PEAK = 157.3


def bench(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8192", "4096"])]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cfgs = laser_amd.f32_configs()
    only = set(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
    results = []
    for n in sizes:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
        B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
        Bt = B.t().contiguous().t()      # transposed storage, same logical matrix
        C = torch.zeros((n, n), device="cuda")
        variants = []
        for ci, name in enumerate(cfgs):
            if only is not None and ci not in only:
                continue
            for mode in (0, 1):
                for lay, (Av, Bv) in (("nn", (A, B)), ("nt", (A, Bt))):
                    variants.append((ci, name, mode, lay, Av, Bv))
        best = {}
        for r in range(rounds + 1):
            for (ci, name, mode, lay, Av, Bv) in variants:
                laser_amd.set_f32_config(ci)
                laser_amd.set_float_mode(mode)
                try:
                    ms = bench(lambda: laser_amd.matmul(Av, Bv, 1, 0, C), 1 if r == 0 else 3)
                except laser_amd.LaserHipError as e:
                    ms = float("inf")
                if r > 0:
                    best.setdefault((ci, mode, lay), []).append(ms)
        for (ci, mode, lay), v in sorted(best.items()):
            v = sorted(v)
            med, mn = v[len(v) // 2], v[0]
            tf = 2.0 * n ** 3 / (med * 1e-3) / 1e12
            rec = {"n": n, "cfg": cfgs[ci], "mode": "laser_order" if mode == 0 else "fast", "layout": lay,
                   "ms_med": round(med, 4), "ms_min": round(mn, 4), "tflops_med": round(tf, 2),
                   "frac_peak": round(tf / PEAK, 4)}
            results.append(rec)
            print(json.dumps(rec), flush=True)
    laser_amd.set_f32_config(-1)
    laser_amd.set_float_mode(0)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "sweep_f32.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(results, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
