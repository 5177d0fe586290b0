#!/usr/bin/env python3
"""Few output tiles x long K: Laser's kc slices as one batched launch + ordered combine vs the sequential K loop
(GPU box).  Results must be identical bit for bit (laser-order) -- checked -- and are timed both ways."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 6)
    ts.sort(); return ts[2]
SHAPES = [(8192, 16, 8192), (8192, 64, 8192), (8192, 128, 8192), (64, 8192, 8192), (512, 512, 131072), (1024, 1024, 8192), (1024, 1024, 65536),
          (256, 256, 4096), (1000, 100, 5000), (1536, 1024, 4100), (32768, 32, 2048)]
for dtype in (torch.float32, torch.float64):
    for (M, N, K) in SHAPES if dtype == torch.float32 else SHAPES[1:6:2]:
        A = (torch.rand((M, K), device="cuda", dtype=dtype) - 0.5) * 0.2; B = (torch.rand((K, N), device="cuda", dtype=dtype) - 0.5) * 0.2
        C0 = (torch.rand((M, N), device="cuda", dtype=dtype) - 0.5)
        out = {}
        for on in (False, True):
            laser_amd.set_slice_parallel(on)
            C = C0.clone()
            laser_amd.matmul(A, B, 0.5, -1.5, C)
            out[on] = C.clone()
            ms = bench(lambda: laser_amd.matmul(A, B, 1, 0, C))
            out[(on, "ms")] = ms
        same = torch.equal(out[False], out[True])
        fl = 2.0 * M * N * K
        print(f"{str(dtype)[6:]} {M}x{N}x{K}: sequential {out[(False,'ms')]:.4f} ms ({fl/out[(False,'ms')]/1e9:.1f} TF)  slice-parallel "
              f"{out[(True,'ms')]:.4f} ms ({fl/out[(True,'ms')]/1e9:.1f} TF)  identical {same}", flush=True)
laser_amd.set_slice_parallel(True)
