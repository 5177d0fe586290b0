#!/usr/bin/env python3
"""Round 6: the any-geometry assembly implicit-GEMM loader against the compiler-scheduled loaders of the same library, per shape and
accumulation mode (f32_asm 1 vs 0; interleaved), same bits required in laser-order mode.  One JSON line per (shape, mode).
usage: conv_geometry_ab.py [batch] [m64]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def t(fn, inner=4, reps=5):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(1)
SHAPES = [((3, 224, 224), (64, 3, 7, 7), (3, 3), (2, 2)), ((128, 56, 56), (256, 128, 3, 3), (1, 1), (2, 2)), ((64, 56, 56), (128, 64, 5, 5), (2, 2), (1, 1)),
          ((256, 28, 28), (512, 256, 1, 1), (0, 0), (1, 1)), ((64, 56, 56), (256, 64, 1, 1), (0, 0), (1, 1)), ((256, 56, 56), (64, 256, 1, 1), (0, 0), (1, 1)),
          ((256, 14, 14), (256, 256, 3, 3), (1, 1), (1, 1)), ((512, 7, 7), (512, 512, 3, 3), (1, 1), (1, 1)), ((128, 28, 28), (128, 128, 3, 3), (1, 1), (1, 1)),
          ((64, 57, 57), (128, 64, 3, 3), (1, 1), (1, 1)), ((32, 112, 112), (64, 32, 3, 3), (1, 1), (2, 2)), ((16, 64, 64), (96, 16, 7, 7), (3, 3), (1, 1))]
if len(sys.argv) > 2 and sys.argv[2] == "m64":      # few output channels: the 64-row assembly tile against the compiler-scheduled kernels
    SHAPES = [((64, 56, 56), (64, 64, 3, 3), (1, 1), (1, 1)), ((128, 56, 56), (64, 128, 3, 3), (1, 1), (1, 1)), ((32, 56, 56), (64, 32, 3, 3), (1, 1), (1, 1)),
              ((64, 28, 28), (64, 64, 3, 3), (1, 1), (1, 1)), ((256, 28, 28), (64, 256, 3, 3), (1, 1), (1, 1)), ((64, 56, 56), (48, 64, 3, 3), (1, 1), (1, 1)),
              ((64, 56, 56), (33, 64, 3, 3), (1, 1), (1, 1)), ((64, 112, 112), (64, 64, 3, 3), (1, 1), (1, 1)), ((128, 56, 56), (64, 128, 1, 1), (0, 0), (1, 1)),
              ((64, 56, 56), (64, 64, 5, 5), (2, 2), (1, 1)), ((96, 35, 35), (64, 96, 3, 3), (1, 1), (1, 1)), ((64, 56, 56), (96, 64, 3, 3), (1, 1), (1, 1))]
for (chw, kshape, pad, st) in SHAPES:
    ishape = (batch,) + chw
    x = torch.rand(ishape, generator=g, device="cuda")
    w = torch.rand(kshape, generator=g, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
    o = torch.zeros(oshape, device="cuda")
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * kshape[2] * kshape[3]
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"ishape": ishape, "kshape": kshape, "pad": pad, "stride": st, "mode": "fast" if mode else "laser_order"}
        outs = {}
        for asm in (1, 0, 1, 0):
            laser_amd.set_f32_asm(asm)
            ms = t(lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, st, None))
            key = "asm" if asm else "compiler"
            if key not in rec or ms < rec[key]["ms"]:
                rec[key] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / 157.3, 3), "kernel": laser_amd.last_f32_asm(),
                            "cut": laser_amd.last_split(), "tail": laser_amd.get_option("last_conv_tail"), "cfg": laser_amd.get_option("last_f32_config")}
            outs[asm] = o.clone()
        rec["bit_identical"] = bool(torch.equal(outs[0], outs[1]))       # (required in laser-order mode; one-chain mode has no order to keep: a different cut's tail may sum its kc slices separately)
        rec["asm_gain_pct"] = round(100.0 * (rec["compiler"]["ms"] / rec["asm"]["ms"] - 1.0), 1)
        print(json.dumps(rec), flush=True)
laser_amd.set_float_mode(0)
laser_amd.set_f32_asm(1)
