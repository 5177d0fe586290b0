#!/usr/bin/env python3
"""Implicit conv: LDS input patch vs per-element gather for the B operand (GPU box): identical results? speed?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
def bench(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 6)
    ts.sort(); return ts[len(ts) // 2]
CASES = [((32, 128, 56, 56), (256, 128, 3, 3), (1, 1), (1, 1)), ((16, 128, 56, 56), (256, 128, 3, 3), (1, 1), (1, 1)),
         ((64, 64, 56, 56), (64, 64, 3, 3), (1, 1), (1, 1)), ((32, 256, 28, 28), (512, 256, 3, 3), (1, 1), (1, 1)),
         ((8, 512, 14, 14), (512, 512, 3, 3), (1, 1), (1, 1)), ((8, 64, 112, 112), (128, 64, 3, 3), (1, 1), (2, 2)),
         ((4, 3, 224, 224), (64, 3, 7, 7), (3, 3), (2, 2)), ((8, 96, 28, 28), (192, 96, 5, 5), (2, 2), (1, 1)),
         ((2, 16, 20, 24), (24, 16, 3, 3), (0, 0), (1, 1))]
for ishape, kshape, pad, st in CASES:
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
    out = torch.zeros(oshape, device="cuda"); ref = torch.zeros(oshape, device="cuda")
    flops = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * kshape[2] * kshape[3]
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        res = {}
        for patch in (False, True):
            laser_amd.set_conv_patch(patch)
            out.fill_(float("nan"))
            ms = bench(lambda: laser_amd.conv2d_im2col(out, oshape, x, ishape, w, kshape, pad, st, None))
            res[patch] = (ms, laser_amd.f32_configs()[laser_amd.last_f32_config()])
            if not patch: ref.copy_(out)
        same = torch.equal(out, ref)
        print(f"{ishape}*{kshape} p{pad} s{st} {'laser' if mode == 0 else 'fast '}: gather {res[False][0]:.4f} ms {flops/res[False][0]/1e9:6.1f} TF | "
              f"patch {res[True][0]:.4f} ms {flops/res[True][0]/1e9:6.1f} TF ({res[True][1]}) | identical {same}")
laser_amd.set_conv_patch(True); laser_amd.set_float_mode(0)
