#!/usr/bin/env python3
"""C4 (and neighbours) with the direct pixel-tail kernel (option conv_tail = 1) against the round-3 tail forms (0): per-call time of the
whole convolution, both accumulation modes, results compared bit for bit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from scripts.bench_configs import ev_time
SHAPES = (((32, 128, 56, 56), (256, 128, 3, 3), (1, 1)), ((32, 128, 56, 56), (256, 128, 3, 3), (0, 0)),
          ((64, 64, 56, 56), (64, 64, 3, 3), (1, 1)), ((32, 128, 28, 28), (128, 128, 3, 3), (1, 1)),
          ((16, 256, 28, 28), (512, 256, 3, 3), (1, 1)))
if len(sys.argv) > 1 and sys.argv[1] == "c4":      # C4 and one more shape whose tail this kernel takes
    SHAPES = (SHAPES[0], ((8, 64, 30, 30), (96, 64, 3, 3), (1, 1)))
for ishape, kshape, pad in SHAPES:
    x = torch.rand(ishape, device="cuda"); w = torch.rand(kshape, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, (1, 1))
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * 9
    rec = {"conv": [ishape, kshape, pad]}
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        outs = {}
        for tail in (0, 1, 0, 1):
            laser_amd.set_option("conv_tail", tail)
            o = torch.zeros(oshape, device="cuda")
            fn = lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, (1, 1), None)
            med, mn = ev_time(fn, iters=9, inner=8, warm=20)
            outs[tail] = o
            key = f"{'laser' if mode == 0 else 'fast'}/conv_tail={tail}"
            rec.setdefault(key, []).append({"ms": round(med, 4), "tflops": round(fl / med / 1e9, 1), "asm": laser_amd.last_f32_asm(),
                                            "cut": laser_amd.last_split(), "tail_form": laser_amd.get_option("last_conv_tail")})
        rec[f"{'laser' if mode == 0 else 'fast'}/same_bits"] = bool(torch.equal(outs[0], outs[1]))
    laser_amd.set_float_mode(0); laser_amd.set_option("conv_tail", 1)
    print(json.dumps(rec), flush=True)
