"""Direct small-channel convolution (conv_small.hip): time per launch and effective HBM rate on the reference's conv bench
geometry and neighbours, direct vs implicit GEMM.  Run on the GPU box: python scripts/probes/conv_direct_probe.py"""
import torch, json, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import laser_amd
def t(fn, inner=8, reps=7):
    t0 = time.time()
    while time.time() - t0 < 0.3: fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]
g = torch.Generator(device="cuda").manual_seed(1)
for ishape, kshape, pad in (((16, 3, 224, 224), (20, 3, 3, 3), (0, 0)), ((64, 3, 224, 224), (20, 3, 3, 3), (0, 0)), ((8, 8, 128, 128), (16, 8, 3, 3), (0, 0)), ((16, 3, 224, 224), (20, 3, 3, 3), (1, 1)),
                            ((16, 3, 224, 224), (32, 3, 3, 3), (1, 1)), ((8, 8, 128, 128), (16, 8, 3, 3), (1, 1)),
                            ((8, 14, 128, 128), (32, 14, 3, 3), (1, 1)), ((8, 3, 224, 224), (8, 3, 7, 7), (3, 3))):
    st = (1, 1)
    x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st); o = torch.zeros(oshape, device="cuda")
    byts = 4.0 * (x.numel() + o.numel())
    row = {"conv": [ishape, kshape, pad]}
    ref = None
    for direct in (1, 2, 0):    # 1 = shipped choice, 2 = without the scalar-filter forms (matrix cores / LDS filter everywhere), 0 = implicit GEMM
        laser_amd.set_option("conv_direct", direct)
        ms = t(lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, st, None))
        name = {1: "direct", 2: "r03_forms", 0: "implicit"}[direct]
        row[name + "_us"] = round(ms * 1e3, 1)
        if direct: row[name + "_TBps"] = round(byts / ms / 1e9, 2)
        if direct == 1: row["cfg"] = laser_amd.get_option("last_f32_config"); ref = o.clone()
        else: row[name + "_same_bits"] = bool(torch.equal(ref, o))
    laser_amd.set_option("conv_direct", 1)
    print(json.dumps(row), flush=True)
