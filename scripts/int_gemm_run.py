#!/usr/bin/env python3
"""A few launches of the integer limb kernels (int64 and int32, n^3) for rocprofv3.  usage: int_gemm_run.py [iters] [n = 4096]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for dt in (torch.int64, torch.int32):
    hi = 2 ** 62 if dt == torch.int64 else 2 ** 30
    A = torch.randint(-hi, hi, (n, n), device="cuda", dtype=dt)
    B = torch.randint(-hi, hi, (n, n), device="cuda", dtype=dt)
    C = torch.zeros((n, n), device="cuda", dtype=dt)
    for _ in range(iters):
        laser_amd.matmul(A, B, 1, 0, C)
    torch.cuda.synchronize()
