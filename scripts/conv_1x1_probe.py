#!/usr/bin/env python3
"""1x1 / stride 1 / no padding convolutions: the reference's GEMM shortcut (conv2d_im2col.nim:121-153: the input already is the [K, N]
matrix; one batched GEMM launch) against the implicit-GEMM convolution kernels (unit walkers) on the same shapes, interleaved; same
bits required.  One JSON line per (shape, mode).  usage: conv_1x1_probe.py [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [((256, 28, 28), 512), ((64, 56, 56), 256), ((256, 56, 56), 64), ((128, 56, 56), 64), ((512, 14, 14), 2048), ((1024, 14, 14), 256), ((64, 112, 112), 64), ((128, 28, 28), 512), ((256, 56, 56), 256)]


def timed(fn, inner):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


g = torch.Generator(device="cuda").manual_seed(5)
for (chw, cout) in SHAPES:
    ishape, kshape = (batch,) + chw, (cout, chw[0], 1, 1)
    x = torch.rand(ishape, generator=g, device="cuda") - 0.5
    w = torch.rand(kshape, generator=g, device="cuda") - 0.5
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, (0, 0), (1, 1))
    out = torch.zeros(oshape, device="cuda")
    fl = 2.0 * batch * cout * chw[1] * chw[2] * chw[0]
    byts = 4.0 * (batch * chw[0] * chw[1] * chw[2] + batch * cout * chw[1] * chw[2] + cout * chw[0])
    fn = lambda: laser_amd.conv2d_im2col(out, oshape, x, ishape, w, kshape, (0, 0), (1, 1), None)
    inner = max(4, min(40, int(3e-3 / max(fl / 100e12, byts / 4e12))))
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"ishape": ishape, "cout": cout, "mode": "fast" if mode else "laser_order", "hbm_floor_us": round(byts / 8e12 * 1e6, 1)}
        outs, ts = {}, {0: [], 1: []}
        for imp in (0, 1):
            laser_amd.set_option("conv_1x1_implicit", imp)
            out.fill_(float("nan")); fn(); torch.cuda.synchronize()
            outs[imp] = out.clone()
            rec["implicit" if imp else "gemm_shortcut"] = {"kernel": laser_amd.last_f32_asm(), "cfg": laser_amd.get_option("last_f32_config"), "wgs": laser_amd.get_option("last_asm_wgs")}
        for _ in range(30):
            fn()
        for _ in range(5):
            for imp in (0, 1):
                laser_amd.set_option("conv_1x1_implicit", imp)
                fn()
                ts[imp].append(timed(fn, inner))
        for imp in (0, 1):
            t = sorted(ts[imp]); ms = t[len(t) // 2]
            rec["implicit" if imp else "gemm_shortcut"].update({"us": round(ms * 1e3, 1), "tflops": round(fl / ms / 1e9, 1), "hbm_frac": round(byts / ms / 1e-3 / 8e12, 3)})
        rec["same_bits"] = bool(torch.equal(outs[0], outs[1]))
        rec["implicit_gain_pct"] = round(100.0 * (rec["gemm_shortcut"]["us"] / rec["implicit"]["us"] - 1.0), 1)
        print(json.dumps(rec), flush=True)
laser_amd.set_option("conv_1x1_implicit", 0)
laser_amd.set_float_mode(0)
