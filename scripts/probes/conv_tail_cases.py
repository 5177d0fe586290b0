import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, laser_amd as la
from oracle import oracle
rng = np.random.default_rng(515)
cases = [((4, 32, 30, 30), (64, 32, 3, 3), (1, 1)), ((3, 96, 26, 26), (50, 96, 3, 3), (0, 0)), ((2, 160, 20, 22), (40, 160, 3, 3), (2, 2)),
         ((5, 64, 57, 57), (33, 64, 3, 3), (1, 1)), ((3, 96, 26, 26), (128, 96, 3, 3), (0, 0)), ((2, 160, 20, 22), (96, 160, 3, 3), (2, 2)), ((5, 64, 57, 57), (65, 64, 3, 3), (1, 1))]
for ishape, kshape, pad in cases:
    x = rng.uniform(0, 1, ishape).astype(np.float32); w = rng.uniform(0, 1, kshape).astype(np.float32)
    oshape = la.conv2d_out_shape(ishape, kshape, pad, (1, 1))
    want = oracle.conv2d_im2col(x, w, pad, (1, 1))
    dx, dw = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    la.set_f32_asm(2); la.set_option("conv_cut_always", 1)
    for tail in (1, 0):
        la.set_option("conv_tail", tail)
        o = torch.full(oshape, float("nan"), device="cuda")
        la.conv2d_im2col(o, oshape, dx, ishape, dw, kshape, pad, (1, 1), None)
        print(ishape, kshape, pad, "tail", tail, "asm", la.last_f32_asm(), "form", la.get_option("last_conv_tail"), "split", la.last_split(), "equal", np.array_equal(o.cpu().numpy(), want), flush=True)
