#!/usr/bin/env python3
"""Randomised convolution parity (GPU box): random geometry (incl. W % 4 == 0 shapes that take the LDS input-patch
loader and others that take the per-element gather), bias + relu epilogue on some, device path, vs the oracle.
usage: fuzz_conv.py [cases] [seed] [small]     small: every case in the direct kernels' class (<= 32 output channels, C*kH*kW <= 256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import laser_amd
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
small = len(sys.argv) > 3 and sys.argv[3] == "small"
direct = 0
tailk = 0
asmk = asm_want = walked = 0
isa = oracle.fused_isa(np.float32)
for it in range(cases):
    kH, kW = int(rng.integers(1, 8)), int(rng.integers(1, 8))
    pH, pW = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    sH, sW = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    H = int(rng.integers(max(kH - 2 * pH, sH + 1, 2), 44))
    W = int(rng.integers(max(kW - 2 * pW, sW + 1, 2), 44))
    if rng.random() < 0.5: W = max(4, (W // 4) * 4)
    if not (sH < H and sW < W and H + 2 * pH >= kH and W + 2 * pW >= kW):
        continue
    n, C, Co = int(rng.integers(1, 4)), int(rng.integers(1, 70)), int(rng.integers(1, 150))
    if rng.random() < 0.08:    # enough tiles for the main + tail launch plan (and its K-slice-parallel tail in laser-order mode)
        kH = kW = 3; pH = pW = int(rng.integers(0, 2)); sH = sW = 1
        H = W = int(rng.choice([28, 30, 54, 56, 58]))
        n, C, Co = int(rng.integers(8, 33)), int(rng.choice([32, 57, 64, 100, 128])), int(rng.choice([128, 192, 256]))
    cut_always = 0
    if not small and rng.random() < 0.12:   # the pixel-tail forms on shapes the launch model would leave in one launch: the cut forced at the
        # last whole 128-pixel tile, channel counts of both classes of the direct tail kernel (multiples of 32: the loop without vector
        # address arithmetic; other multiples of 4: the tap table), tails of every length, blocks breaking across image rows, padding 0..2
        kH = kW = 3; pH = pW = int(rng.integers(0, 3)); sH = sW = 1
        H, W = int(rng.integers(12, 40)), int(rng.choice([12, 14, 16, 20, 22, 24, 26, 28, 30, 34, 36, 38]))
        n, C, Co = int(rng.integers(1, 6)), int(rng.choice([4, 8, 20, 32, 60, 64, 96, 100, 128, 160])), int(rng.integers(1, 200))
        cut_always = 1
    asm_any, walk = 0, 1
    if not small and not cut_always and rng.random() < 0.3:
        # round 6: the assembly loader on ANY geometry (kernel of up to 49 taps, strides, odd widths, any C_in): forced onto the
        # hand-scheduled kernels whatever the tile count, enough output channels to stay out of the direct kernels' class
        asm_any = 1
        Co = int(rng.choice([33, 40, 64, 100, 128, 200, 256, 300]))
        C = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 23, 32, 40, 64]))
        if rng.random() < 0.3: H, W = int(rng.integers(max(kH, 30), 90)), int(rng.integers(max(kW, 30), 90))
        if rng.random() < 0.5:
            # the unit-walking form of the kernels (pipelined transitions need K a multiple of 32, three K-tiles or more, the plain
            # epilogue): forced onto few workgroups, so that every one walks several units (2 = one per slot)
            C = int(rng.choice([32, 64, 96])); n = int(rng.integers(2, 5)); walk = int(rng.choice([2, 3, 4, 5, 7]))
    if small:
        Co = int(rng.integers(1, 33))
        C = int(rng.integers(1, max(2, min(70, 256 // (kH * kW) + 1))))
        if rng.random() < 0.3: H, W = int(rng.integers(max(kH, 40), 120)), int(rng.integers(max(kW, 40), 120))   # several groups per wave
    ishape, kshape, pad, st = (n, C, H, W), (Co, C, kH, kW), (pH, pW), (sH, sW)
    x = rng.uniform(-1, 1, ishape).astype(np.float32); w = rng.uniform(-1, 1, kshape).astype(np.float32)
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
    want = oracle.conv2d_im2col(x, w, pad, st, isa=isa)
    # the reference's 1x1 shortcut (conv2d_im2col.nim:124-153, restated by the oracle) treats the input as the
    # K x N matrix, which is only right for stride 1 / no padding; the product does a real convolution there
    # (superset 2), so those cases are checked against the direct convolution instead
    shortcut_is_wrong = kH * kW == 1 and (pad != (0, 0) or st != (1, 1))
    if shortcut_is_wrong:
        want = oracle.conv2d_direct(x, w, pad, st)
    use_epi = rng.random() < 0.3
    b = rng.uniform(-1, 1, Co).astype(np.float32) if use_epi else None
    if use_epi: want = oracle.apply_epilogue(want, b.reshape(1, -1, 1, 1), "relu")
    dout = torch.full(oshape, float("nan"), device="cuda")
    laser_amd.set_conv_patch(bool(rng.random() < 0.8)); laser_amd.set_conv_kslice(bool(rng.random() < 0.8))
    laser_amd.set_option("conv_tail", int(rng.random() < 0.7))      # the direct pixel-tail kernel / the round-3 tail forms
    laser_amd.set_option("conv_walk", walk)
    laser_amd.set_option("conv_cut_always", cut_always); laser_amd.set_f32_asm(2 if (cut_always or asm_any) else 1)
    laser_amd.conv2d_im2col(dout, oshape, torch.from_numpy(x).cuda(), ishape, torch.from_numpy(w).cuda(), kshape, pad, st, None,
                            bias=None if b is None else torch.from_numpy(b).cuda(), activation="relu" if use_epi else None)
    direct += laser_amd.get_option("last_f32_config") == -3
    tailk += laser_amd.get_option("last_conv_tail") == 1
    asmk += laser_amd.last_f32_asm() != 0
    asm_want += asm_any
    walked += laser_amd.last_f32_asm() >= 67
    if asm_any and kH * kW <= 49 and laser_amd.last_f32_asm() == 0 and laser_amd.get_option("last_f32_config") != -3:
        fails += 1
        print("FAIL (not on the assembly loader)", dict(it=it, ishape=ishape, kshape=kshape, pad=pad, st=st), flush=True)
    got = dout.cpu().numpy()
    ok = np.allclose(got, want, rtol=1e-5, atol=1e-5) if shortcut_is_wrong else np.array_equal(got, want)
    if not ok:
        fails += 1
        print("FAIL", dict(it=it, ishape=ishape, kshape=kshape, pad=pad, st=st, epi=use_epi, maxabs=float(np.nanmax(np.abs(got - want)))), flush=True)
    # the explicit im2col of the same geometry (the public im2col*[T] entry: band kernel, float32 / float64, device-resident, batched,
    # a random band length now and then): pure data movement, bit-exact against the oracle's im2col of every image
    if it % 3 == 0 and oshape[2] * oshape[3] * C * kH * kW * n < (1 << 24):
        import ctypes
        f64 = bool(rng.random() < 0.4)
        dt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
        xi = np.rint(x * 100).astype(np.float32)          # integer-valued: the float32 oracle pins the float64 path too
        wsw = np.stack([oracle.im2col(xi[i], kshape, pad, st) for i in range(n)]).astype(dt)
        d_in = torch.from_numpy(xi.astype(dt)).cuda()
        d_ws = torch.full(wsw.shape, -7, dtype=tdt, device="cuda")
        laser_amd.set_option("im2col_band", int(rng.choice([0, 0, 4, 64, 300, 5000])))
        fn = laser_amd.lib().laser_hip_im2col_f64_dev if f64 else laser_amd.lib().laser_hip_im2col_f32_dev
        rc = fn(ctypes.c_void_p(d_ws.data_ptr()), oshape[2], oshape[3], ctypes.c_void_p(d_in.data_ptr()), n, C, H, W, kH, kW, pH, pW, sH, sW,
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        laser_amd.set_option("im2col_band", 0)
        if rc != 0 or not np.array_equal(d_ws.cpu().numpy(), wsw):
            fails += 1
            print("FAIL im2col", dict(it=it, ishape=ishape, kshape=kshape, pad=pad, st=st, f64=f64, rc=rc), flush=True)
laser_amd.set_conv_patch(True); laser_amd.set_conv_kslice(True); laser_amd.set_option("conv_tail", 1)
laser_amd.set_option("conv_cut_always", 0); laser_amd.set_option("conv_walk", 1); laser_amd.set_f32_asm(1)
print(f"fuzz_conv: {cases} cases, {fails} failures, {direct} on the direct small-channel kernels, {tailk} with the direct pixel-tail kernel, "
      f"{asmk} on the assembly implicit-GEMM loader ({asm_want} forced there with a random kernel / stride / width; {walked} as unit walkers with pipelined transitions)")
sys.exit(1 if fails else 0)
