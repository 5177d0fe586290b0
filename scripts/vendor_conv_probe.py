#!/usr/bin/env python3
"""Context only (nothing in the product calls a vendor library): torch.nn.functional.conv2d -> MIOpen on the same box, the same
operands and timing harness as bench_configs.py's C4 lines, next to laser_hip's two accumulation modes.  MIOpen is free to use
algorithms that change the arithmetic (Winograd, FFT); laser_hip's laser-order mode is bit-identical to Laser's im2col + GEMM."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from scripts.bench_configs import ev_time
torch.backends.cudnn.allow_tf32 = False
for ishape, kshape, pad in (((32, 128, 56, 56), (256, 128, 3, 3), (1, 1)), ((32, 128, 56, 56), (256, 128, 3, 3), (0, 0)),
                            ((16, 3, 224, 224), (20, 3, 3, 3), (0, 0))):
    x = torch.rand(ishape, device="cuda"); w = torch.rand(kshape, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, (1, 1))
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * kshape[2] * kshape[3]
    rec = {"conv": [ishape, kshape, pad]}
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        fn = lambda: torch.nn.functional.conv2d(x, w, None, 1, pad)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        med, mn = ev_time(fn, iters=9, inner=8, warm=20)
        rec[f"vendor_conv_ms{'_autotuned' if bench else ''}"] = round(med, 4); rec[f"vendor_conv_tflops{'_autotuned' if bench else ''}"] = round(fl / med / 1e9, 1)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, pad)
    for mode, name in ((0, "laser_order"), (1, "fast")):
        laser_amd.set_float_mode(mode)
        o = torch.zeros(oshape, device="cuda")
        f2 = lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, (1, 1), None)
        med, mn = ev_time(f2, iters=9, inner=8, warm=20)
        rec[f"laser_hip_{name}_ms"] = round(med, 4); rec[f"laser_hip_{name}_tflops"] = round(fl / med / 1e9, 1)
        rec[f"laser_hip_{name}_max_rel_err_vs_f64"] = float(((o.double() - ref).abs().max() / ref.abs().max()).item())
    laser_amd.set_float_mode(0)
    v = torch.nn.functional.conv2d(x, w, None, 1, pad)
    rec["vendor_conv_max_rel_err_vs_f64"] = float(((v.double() - ref).abs().max() / ref.abs().max()).item())
    print(json.dumps(rec), flush=True)
