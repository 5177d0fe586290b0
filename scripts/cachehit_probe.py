#!/usr/bin/env python3
"""Is the f32 kernel's remaining gap memory-side?  Time it on operands that are broadcast views
(row stride 0): identical instruction stream, every load an L1/L2 hit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
n = 8192
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
Ab = A[:1].expand(n, n)   # strides (0, 1)
Bb = B[:1].expand(n, n)   # strides (0, 1)
C = torch.zeros((n, n), device="cuda")
def bench(fn, iters=4):
    fn(); fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    ts.sort(); return ts[len(ts) // 2]
names = laser_amd.f32_configs()
for cfg in range(len(names)):
    for mode in (0, 1):
        laser_amd.set_f32_config(cfg); laser_amd.set_float_mode(mode)
        t_real = bench(lambda: laser_amd.matmul(A, B, 1, 0, C))
        t_hit = bench(lambda: laser_amd.matmul(Ab, Bb, 1, 0, C))
        print(f"{names[cfg]:24s} {'laser' if mode == 0 else 'fast ':5s} real {2*n**3/t_real/1e9:7.1f} TF   all-cache-hit {2*n**3/t_hit/1e9:7.1f} TF")
