#!/usr/bin/env python3
"""Round split (capi.cpp run_gemm_round_split: whole rounds of tiles on top + the K-sliced rows of the badly filled last round)
vs the single launch (option split_tail = 0): TFLOP/s and bit-identity per shape, laser-order mode.  One JSON line per shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd


def t_ms(fn, inner=6, reps=5):
    t0 = time.time()
    while time.time() - t0 < 0.15: fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(3)
shapes = [("f32", s) for s in [(1536, 1536, 1536), (1792, 1792, 1792), (2304, 2304, 2304), (2560, 2560, 2560), (1280, 1280, 1280), (1920, 1920, 1920),
                               (1536, 1536, 4096), (1100, 1700, 2048), (3072, 3072, 3072), (2048, 2048, 2048), (1664, 1664, 1024), (4100, 4100, 4100)]]
shapes += [("f64", s) for s in [(1536, 1536, 1536), (1792, 1792, 1792), (2304, 2304, 2304), (960, 960, 960), (1280, 1280, 1280), (3072, 3072, 3072)]]
for dt, (M, N, K) in shapes:
    tdt = torch.float32 if dt == "f32" else torch.float64
    A = torch.rand((M, K), generator=g, device="cuda", dtype=tdt) - 0.5
    B = torch.rand((K, N), generator=g, device="cuda", dtype=tdt) - 0.5
    rec = {"dtype": dt, "shape": [M, N, K]}
    outs = {}
    for split in (1, 0):
        laser_amd.set_option("split_tail", split)
        C = torch.zeros((M, N), device="cuda", dtype=tdt)
        ms = t_ms(lambda: laser_amd.matmul(A, B, 1, 0, C))
        key = "split" if split else "single"
        rec[key + "_tflops"] = round(2.0 * M * N * K / ms / 1e9, 1)
        if split: rec["row_cut"] = -laser_amd.get_option("last_split") if laser_amd.get_option("last_split") < 0 else 0
        outs[split] = C
    laser_amd.set_option("split_tail", 1)
    rec["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
    print(json.dumps(rec), flush=True)
