#!/usr/bin/env python3
"""BASELINE configs[3] (conv N=32 C=128 H=W=56 K=256 R=S=3, pad 1, stride 1) through the library's default
implicit-GEMM path (hand-scheduled main launch + compiler-scheduled pixel tail), laser-order then fast, for timing
and for rocprofv3 (scripts/gpu_profile_cmd.sh conv_c4 python scripts/conv_c4_run.py).   usage: conv_c4_run.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ishape, kshape, pad, st = (32, 128, 56, 56), (256, 128, 3, 3), (1, 1), (1, 1)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
out = torch.zeros(oshape, device="cuda")
flops = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * 9
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    fn = lambda: laser_amd.conv2d_im2col(out, oshape, x, ishape, w, kshape, pad, st, None)
    for _ in range(60): fn()      # (the clocks ramp up over the first ~50 launches after idle: 470 -> 405 us per main launch)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    ts.sort()
    print(f"{'laser' if mode == 0 else 'fast '} asm_kernel={laser_amd.last_f32_asm()} cut={laser_amd.last_split()} "
          f"{ts[2]:.4f} ms (min {ts[0]:.4f}) {flops/ts[2]/1e9:6.1f} TF", flush=True)
laser_amd.set_float_mode(0)
