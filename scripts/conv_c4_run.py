#!/usr/bin/env python3
"""BASELINE configs[3] (conv N=32 C=128 H=W=56 K=256 R=S=3, pad 1, stride 1) through the implicit-GEMM path, for
timing and for rocprofv3: runs the LDS-patch loader and the per-element gather loader (distinct kernel template
arguments, so one profile separates them), laser-order and fast, main + tail cut on and off.
usage: conv_c4_run.py [iters]"""
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
names = laser_amd.f32_configs()
ref = None
for patch in (1, 0):          # LDS input patch / per-element gather
    for mode in (0, 1):
        for split, ks in ((1, 1), (1, 0), (0, 1)):   # main + tail with the K-slice-parallel tail / sequential tail; one launch
            if mode == 1 and ks == 0:
                continue                              # (the K-slice tail is a laser-order mechanism)
            laser_amd.set_conv_patch(patch); laser_amd.set_float_mode(mode); laser_amd.set_split_tail(split); laser_amd.set_conv_kslice(ks)
            fn = lambda: laser_amd.conv2d_im2col(out, oshape, x, ishape, w, kshape, pad, st, None)
            for _ in range(3): fn()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): fn()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
            ts.sort()
            if mode == 0:
                if ref is None: ref = out.clone()
                assert torch.equal(ref, out), "laser-order result depends on the launch plan / loader"
            print(f"loader={('gather', 'patch ')[patch]} {'laser' if mode == 0 else 'fast '} split={split} kslice={ks} "
                  f"cfg={names[laser_amd.last_f32_config()]} cut={laser_amd.last_split()} "
                  f"{ts[2]:.4f} ms (min {ts[0]:.4f}) {flops/ts[2]/1e9:6.1f} TF", flush=True)
laser_amd.set_conv_patch(1); laser_amd.set_float_mode(0); laser_amd.set_split_tail(1); laser_amd.set_conv_kslice(1)
