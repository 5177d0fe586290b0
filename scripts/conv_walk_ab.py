#!/usr/bin/env python3
"""Round 6: the convolution main launch as unit walkers with pipelined transitions (option conv_walk; f32_kernel.py Cfg.cpers) against
the one-tile-per-workgroup launch of the same kernels, per shape and accumulation mode, interleaved; same bits required.  One JSON line
per (shape, mode).  usage: conv_walk_ab.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = [  # (input NCHW, filter, pad, stride)
    ((32, 128, 56, 56), (256, 128, 3, 3), (1, 1), (1, 1)),      # BASELINE configs[3]
    ((32, 128, 56, 56), (256, 128, 3, 3), (0, 0), (1, 1)),
    ((64, 64, 56, 56), (64, 64, 3, 3), (1, 1), (1, 1)),
    ((16, 256, 28, 28), (512, 256, 3, 3), (1, 1), (1, 1)),
    ((32, 128, 28, 28), (128, 128, 3, 3), (1, 1), (1, 1)),
    ((32, 64, 56, 56), (128, 64, 5, 5), (2, 2), (1, 1)),
    ((32, 128, 56, 56), (256, 128, 3, 3), (1, 1), (2, 2)),
    ((64, 256, 28, 28), (256, 256, 3, 3), (1, 1), (1, 1)),
    ((128, 64, 56, 56), (128, 64, 3, 3), (1, 1), (1, 1)),
]


def timed(fn, inner):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


g = torch.Generator(device="cuda").manual_seed(3)
for ishape, kshape, pad, st in SHAPES:
    x = torch.rand(ishape, generator=g, device="cuda") - 0.5
    w = torch.rand(kshape, generator=g, device="cuda") - 0.5
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
    out = torch.zeros(oshape, device="cuda")
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * kshape[2] * kshape[3]
    fn = lambda: laser_amd.conv2d_im2col(out, oshape, x, ishape, w, kshape, pad, st, None)
    inner = max(4, min(40, int(4e-3 / (fl / 120e12))))
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"ishape": ishape, "kshape": kshape, "pad": pad, "stride": st, "mode": "fast" if mode else "laser_order"}
        outs, ts = {}, {0: [], 1: []}
        for walk in (0, 1):
            laser_amd.set_option("conv_walk", walk)
            out.fill_(float("nan"))
            fn()
            torch.cuda.synchronize()
            outs[walk] = out.clone()
            rec["walk" if walk else "plain"] = {"kernel": laser_amd.last_f32_asm(), "wgs": laser_amd.get_option("last_asm_wgs"), "cut": laser_amd.last_split()}
        for _ in range(50):
            fn()
        for _ in range(reps):
            for walk in (0, 1):
                laser_amd.set_option("conv_walk", walk)
                fn()
                ts[walk].append(timed(fn, inner))
        for walk in (0, 1):
            t = sorted(ts[walk])
            rec["walk" if walk else "plain"].update({"ms": round(t[len(t) // 2], 4), "min_ms": round(t[0], 4), "tflops": round(fl / t[len(t) // 2] / 1e9, 1),
                                                     "frac": round(fl / t[len(t) // 2] / 1e9 / 157.3, 4)})
        rec["same_bits"] = bool(torch.equal(outs[0], outs[1]))
        rec["gain_pct"] = round(100.0 * (rec["plain"]["ms"] / rec["walk"]["ms"] - 1.0), 2)
        print(json.dumps(rec), flush=True)
laser_amd.set_option("conv_walk", 1)
laser_amd.set_float_mode(0)
