#!/usr/bin/env python3
"""A few launches of the float64 MFMA kernel (4096^3, laser-order and fast) for rocprofv3.  usage: f64_gemm_run.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 4096
A = (torch.rand((n, n), device="cuda", dtype=torch.float64) - 0.5) * 0.2
B = (torch.rand((n, n), device="cuda", dtype=torch.float64) - 0.5) * 0.2
C = torch.zeros((n, n), device="cuda", dtype=torch.float64)
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    for _ in range(iters):
        laser_amd.matmul(A, B, 1, 0, C)
    torch.cuda.synchronize()
laser_amd.set_float_mode(0)
