#!/usr/bin/env python3
"""Where does the alpha != 1 / beta != 0 variant of the 8192^3 sgemm lose its 3 %? (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
n = 8192
A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2
C = (torch.rand((n, n), device="cuda") - 0.5) * 0.2
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    ts.sort(); return ts[2]
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    for (al, be) in [(1, 0), (0.5, 0), (1, 1), (1, 0.25), (0.5, 0.25), (1, 0)]:
        ms = bench(lambda: laser_amd.matmul(A, B, al, be, C))
        C.uniform_(-0.1, 0.1)
        print("laser" if mode == 0 else "fast ", f"alpha={al} beta={be}: {ms:.4f} ms  {2*n**3/ms/1e9:.1f} TF", flush=True)
