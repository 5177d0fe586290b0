#!/usr/bin/env python3
"""Why is the conv-shaped GEMM slow?  Separate batching, N raggedness and short K (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
def bench(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    ts.sort(); return ts[2]
def rnd(*shape): return (torch.rand(shape, device="cuda") - 0.5) * 0.2
cases = []
K = 1152
A = rnd(256, K)
B32 = rnd(32, K, 3136); C32 = torch.zeros(32, 256, 3136, device="cuda")
cases.append(("batched 32 x (256 x 3136 x 1152)", 2.0 * 32 * 256 * 3136 * K,
              lambda: laser_amd.gemm_strided_batched(32, 256, 3136, K, 1.0, A, K, 1, 0, B32, 3136, 1, K * 3136, 0.0, C32, 3136, 1, 256 * 3136)))
Bw = rnd(K, 100352); Cw = torch.zeros(256, 100352, device="cuda")
cases.append(("single 256 x 100352 x 1152", 2.0 * 256 * 100352 * K, lambda: laser_amd.matmul(A, Bw, 1, 0, Cw)))
B3 = rnd(32, K, 3072); C3 = torch.zeros(32, 256, 3072, device="cuda")
cases.append(("batched 32 x (256 x 3072 x 1152) [N multiple of 256]", 2.0 * 32 * 256 * 3072 * K,
              lambda: laser_amd.gemm_strided_batched(32, 256, 3072, K, 1.0, A, K, 1, 0, B3, 3072, 1, K * 3072, 0.0, C3, 3072, 1, 256 * 3072)))
A8 = rnd(8192, K); B8 = rnd(K, 8192); C8 = torch.zeros(8192, 8192, device="cuda")
cases.append(("8192 x 8192 x 1152 (short K only)", 2.0 * 8192 * 8192 * K, lambda: laser_amd.matmul(A8, B8, 1, 0, C8)))
A2 = rnd(2048, K); B2 = rnd(K, 12544); C2 = torch.zeros(2048, 12544, device="cuda")
cases.append(("2048 x 12544 x 1152 (same flops as C4, squarer)", 2.0 * 2048 * 12544 * K, lambda: laser_amd.matmul(A2, B2, 1, 0, C2)))
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    for name, fl, fn in cases:
        ms = bench(fn)
        print(f"{'laser' if mode == 0 else 'fast '} {name:55s} {ms:.4f} ms {fl/ms/1e9:6.1f} TF")
