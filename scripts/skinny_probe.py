#!/usr/bin/env python3
"""Matrix-vector-like shapes: streaming kernel vs the tiled kernels (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 8)
    ts.sort(); return ts[3]
for (M, N, K, tb) in [(8192, 1, 8192, False), (1, 8192, 8192, False), (8192, 4, 8192, False), (8, 16384, 4096, False), (65536, 1, 4096, False),
                      (1, 8192, 8192, True), (16384, 8, 16384, False)]:
    A = (torch.rand((M, K), device="cuda") - 0.5) * 0.2
    B = (torch.rand((N, K), device="cuda") - 0.5).t() * 0.2 if tb else (torch.rand((K, N), device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    byts = 4.0 * (M * K + K * N + M * N)
    res = []
    for on in (False, True):
        laser_amd.set_skinny(on)
        ms = bench(lambda: laser_amd.matmul(A, B, 1, 0, C))
        res.append(ms)
    print(f"{M}x{N}x{K}{' (B transposed)' if tb else ''}: tiled {res[0]:.4f} ms ({byts/res[0]/1e9:.2f} TB/s)  streaming {res[1]:.4f} ms ({byts/res[1]/1e9:.2f} TB/s)", flush=True)
laser_amd.set_skinny(True)
