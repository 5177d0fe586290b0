#!/usr/bin/env python3
"""C4 convolution under every f32 tile configuration / accumulation mode (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
ishape, kshape, pad, st = (32, 128, 56, 56), (256, 128, 3, 3), (1, 1), (1, 1)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
out = torch.zeros(oshape, device="cuda")
flops = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * 9
def bench(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    ts.sort(); return ts[2]
for cfg, name in enumerate(laser_amd.f32_configs()):
    for mode in (0, 1):
        laser_amd.set_f32_config(cfg); laser_amd.set_float_mode(mode)
        ms = bench(lambda: laser_amd.conv2d_im2col(out, oshape, x, ishape, w, kshape, pad, st, None))
        print(f"{name:24s} {'laser' if mode == 0 else 'fast ':5s} {ms:.4f} ms {flops/ms/1e9:6.1f} TF")
