#!/usr/bin/env python3
"""transpose2D_copy under every tile-shape / streaming-hint variant (GPU box); also a plain D2D copy as the ceiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from laser_amd import _lib
L = _lib.lib()
NAMES = {0: "64r x128c (production)", 1: "64x64", 2: "64x64 nt", 3: "64r x128c nt", 4: "128r x64c", 5: "128x128",
         6: "32r x64c", 7: "32r x128c"}
def bench(fn, reps=6):
    for _ in range(3): fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    ts.sort(); return ts[len(ts) // 2]
for (NR, NC) in [(16384, 8192), (8192, 8192), (4096, 4096), (4000, 2000), (20000, 10000)]:
    src = torch.rand((NR, NC), device="cuda"); dst = torch.empty((NC, NR), device="cuda")
    gb = 2 * NR * NC * 4 / 1e9
    ms = bench(lambda: dst.view(-1).copy_(src.view(-1)))
    print(f"{NR}x{NC}: torch D2D copy           {ms:.4f} ms {gb/ms*1e3/1e3:7.2f} TB/s")
    for v in sorted(NAMES):
        L.laser_hip_set_transpose_variant(v)
        ms = bench(lambda: laser_amd.transpose2D_copy(dst, src, NR, NC))
        ok = torch.equal(dst, src.t())
        print(f"{NR}x{NC}: variant {v} {NAMES[v]:22s} {ms:.4f} ms {gb/ms*1e3/1e3:7.2f} TB/s {'ok' if ok else 'WRONG'}")
    L.laser_hip_set_transpose_variant(0)
