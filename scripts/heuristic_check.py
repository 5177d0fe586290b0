#!/usr/bin/env python3
"""Does the tile heuristic pick (close to) the fastest configuration?  For each shape and accumulation mode:
time the automatic choice and every forced configuration (GPU box).  One JSON line per shape/mode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd

SHAPES = [(512,) * 3, (768,) * 3, (1024,) * 3, (1536,) * 3, (2048,) * 3, (2560,) * 3, (3072,) * 3, (4096,) * 3,
          (4100,) * 3, (6144,) * 3, (8192,) * 3, (1000, 3000, 2000), (256, 100352, 1152), (8192, 512, 8192),
          (512, 8192, 4096), (3000, 3000, 600), (16384, 1024, 1024)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")]


def bench(fn, flop):
    reps = max(2, min(40, int(2e-3 / max(flop / 100e12, 1e-6))))
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02: fn()      # ramp the clocks back up after host-side work
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    ts.sort(); return ts[len(ts) // 2]


names = laser_amd.f32_configs()
for (M, N, K) in SHAPES:
    A = (torch.rand((M, K), device="cuda") - 0.5) * 0.2
    B = (torch.rand((K, N), device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    flop = 2.0 * M * N * K
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        times = {}
        for cfg in [-1] + list(range(len(names))):
            laser_amd.set_f32_config(cfg)
            times[cfg] = bench(lambda: laser_amd.matmul(A, B, 1, 0, C), flop)
            if cfg == -1:
                chosen, cut = laser_amd.last_f32_config(), laser_amd.last_split()
                laser_amd.set_split_tail(0)
                nosplit = bench(lambda: laser_amd.matmul(A, B, 1, 0, C), flop)
                laser_amd.set_split_tail(1)
        best = min((c for c in times if c >= 0), key=lambda c: times[c])
        print(json.dumps({"shape": [M, N, K], "mode": "laser_order" if mode == 0 else "fast", "chosen": names[chosen],
                          "best": names[best], "cut": cut, "auto_ms": round(times[-1], 4), "auto_nosplit_ms": round(nosplit, 4), "best_ms": round(times[best], 4),
                          "auto_tflops": round(flop / times[-1] / 1e9, 1), "auto_over_best": round(times[-1] / times[best], 3),
                          "all_ms": {names[c]: round(t, 4) for c, t in times.items() if c >= 0}}), flush=True)
laser_amd.set_f32_config(-1); laser_amd.set_float_mode(0)
