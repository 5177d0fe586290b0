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
laser_amd.set_split_tail(1)
for cin in (128, 256, 512):
  for pad in ((1,1),(0,0)):
    ishape, kshape, st = (32, cin, 50 if pad[0]==0 else 48, 66 if pad[0]==0 else 64), (256, cin, 3, 3), (1, 1)   # 48x64 = 3072 px = 24 tiles exactly
    x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st); o = torch.zeros(oshape, device="cuda")
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * 9
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        ms = t(lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, st, None))
        print(json.dumps({"conv": ishape, "pad": pad, "oshape": list(oshape), "mode": mode, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "kernel": laser_amd.last_f32_asm(), "cut": laser_amd.last_split()}), flush=True)
    # the same shape as a plain GEMM: M=256, N=32*3072, K=cin*9
    if pad[0] == 1:
      A = torch.rand((256, cin * 9), generator=g, device="cuda"); B = torch.rand((cin * 9, 32 * 3072), generator=g, device="cuda"); C = torch.zeros((256, 32 * 3072), device="cuda")
      for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        ms = t(lambda: laser_amd.matmul(A, B, out=C))
        print(json.dumps({"gemm": [256, 32 * 3072, cin * 9], "mode": mode, "ms": round(ms, 4), "tflops": round(2.0 * 256 * 32 * 3072 * cin * 9 / ms / 1e9, 1), "kernel": laser_amd.last_f32_asm()}), flush=True)
laser_amd.set_float_mode(0)
