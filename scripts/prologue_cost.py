#!/usr/bin/env python3
"""Cost of the fused prologue: 4096^3 plain vs relu(A) in the staging registers vs relu(A) relu(B) vs relu(A) as a
separate elementwise pass + the plain product; and a short-K shape where the pass weighs more.  One JSON line per shape and mode."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd


def t(fn, reps=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (M, N, K) in ((4096, 4096, 4096), (8192, 512, 8192)):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2
    B = (torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    R = torch.empty_like(A)
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"M": M, "N": N, "K": K, "mode": "laser_order" if mode == 0 else "fast"}
        rec["plain_ms"] = round(t(lambda: laser_amd.matmul(A, B, 1, 0, C)), 4)
        rec["fused_relu_a_ms"] = round(t(lambda: laser_amd.matmul(A, B, 1, 0, C, pre=laser_amd.PRE_RELU_A)), 4)
        rec["kernel"] = laser_amd.last_f32_asm()
        rec["fused_relu_ab_ms"] = round(t(lambda: laser_amd.matmul(A, B, 1, 0, C, pre=laser_amd.PRE_RELU_A | laser_amd.PRE_RELU_B)), 4)
        rec["separate_pass_relu_a_ms"] = round(t(lambda: (torch.clamp(A, min=0, out=R), laser_amd.matmul(R, B, 1, 0, C))), 4)
        print(json.dumps(rec), flush=True)
laser_amd.set_float_mode(0)
