import torch, json, sys
sys.path.insert(0, "/root/repo")
import laser_amd
def t(fn, inner=4, reps=5):
    for _ in range(40): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]
g = torch.Generator(device="cuda").manual_seed(1)
for (ishape, kshape, pad) in [((32, 128, 56, 56), (256, 128, 3, 3), (1, 1)), ((32, 128, 56, 56), (256, 128, 3, 3), (0, 0)), ((64, 64, 56, 56), (64, 64, 3, 3), (1, 1)), ((16, 256, 28, 28), (512, 256, 3, 3), (1, 1)), ((32, 128, 28, 28), (128, 128, 3, 3), (1, 1)), ((16, 3, 224, 224), (20, 3, 3, 3), (0, 0))]:
    st = (1, 1)
    x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st); o = torch.zeros(oshape, device="cuda")
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * 9
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"ishape": ishape, "kshape": kshape, "pad": pad, "mode": "fast" if mode else "laser_order"}
        outs = {}
        for asm in (1, 0):      # (the last shape is the direct small-channel kernel's class: "asm" = conv_direct there)
            laser_amd.set_f32_asm(asm); laser_amd.set_option("conv_direct", asm)
            ms = t(lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, st, None))
            rec["asm" if asm else "compiler"] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "kernel": laser_amd.last_f32_asm(), "cut": laser_amd.last_split()}
            outs[asm] = o.clone()
        rec["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
        print(json.dumps(rec), flush=True)
laser_amd.set_float_mode(0); laser_amd.set_f32_asm(1); laser_amd.set_option("conv_direct", 1)
