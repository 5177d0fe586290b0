#!/usr/bin/env python3
"""Randomised parity stress (GPU box): random shapes / layouts / element offsets / leading-dimension padding /
alpha, beta / tile configuration, device path.  float32 + float64 in laser-order mode must equal the oracle bit for
bit; FAST mode must stay within 1e-5 mean relative error.  usage: fuzz_gemm.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import laser_amd
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ncfg = len(laser_amd.f32_configs())
fails = 0
for it in range(cases):
    dtype = [np.float32, np.float64, np.int32, np.int64][int(rng.choice(4, p=[0.55, 0.15, 0.2, 0.1]))]
    is_int = np.dtype(dtype).kind == "i"
    big = rng.random() < 0.25
    M, N, K = (int(rng.integers(1, 1400 if big else 300)) for _ in range(3))
    if rng.random() < 0.3: K = int(rng.integers(500, 1300))      # several kc slices
    shape_kind = rng.random()
    if shape_kind < 0.12:      # matrix-vector-like: the streaming (skinny) kernel
        if rng.random() < 0.5: M, N = int(rng.integers(512, 3000)), int(rng.integers(1, 9))
        else: M, N = int(rng.integers(1, 9)), int(rng.integers(512, 3000))
        K = int(rng.integers(300, 2500))
    elif shape_kind < 0.24:    # few tiles x long K: the slice-parallel form
        M, N, K = int(rng.integers(64, 400)), int(rng.integers(64, 400)), int(rng.integers(1100, 5000))
    elif shape_kind < 0.34:    # small: the one-wave-per-block kernel
        M, N, K = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 129))
    elif shape_kind < 0.42:    # 1..8 rows / columns past a multiple of 64: the peeled launch plan
        M = 64 * int(rng.integers(16, 28)) + (int(rng.integers(1, 9)) if rng.random() < 0.7 else 0)
        N = 64 * int(rng.integers(16, 28)) + (int(rng.integers(1, 9)) if rng.random() < 0.7 else 0)
        K = int(rng.integers(256, 1400))
    elif shape_kind < 0.60 and not is_int:   # many tiles x several kc slices: the assembly kernels' launch plans (below)
        M, N, K = int(rng.integers(300, 1500)), int(rng.integers(300, 1500)), int(rng.integers(520, 3000))
    elif shape_kind < 0.63 and is_int:   # integer K past the limb kernels' 8192-k launch chunk / fold interval
        M, N, K = int(rng.integers(64, 200)), int(rng.integers(64, 200)), int(rng.integers(8193, 17000))
    ta, tb = rng.random() < 0.4, rng.random() < 0.4
    offA, offB, offC = (int(rng.integers(0, 4)) for _ in range(3))
    padA, padB, padC = (int(rng.integers(0, 6)) * int(rng.random() < 0.6) for _ in range(3))
    alpha, beta = (dtype(rng.choice([1, 1, 0, -3, 2] if is_int else [1.0, 1.0, 0.0, -0.5, 2.0])) for _ in range(2))
    def draw(n):
        if is_int:
            info = np.iinfo(dtype)
            return rng.integers(info.min, info.max, n, dtype=dtype)   # full range: wrap-around arithmetic
        return rng.uniform(-0.5, 0.5, n).astype(dtype)
    def make(rows, cols, trans, off, pad):
        r, c = (cols, rows) if trans else (rows, cols)
        ld = c + pad
        buf = draw(off + r * ld + 8)
        view = np.lib.stride_tricks.as_strided(buf[off:], (r, c), (ld * buf.itemsize, buf.itemsize))
        dbuf = torch.from_numpy(buf).cuda()
        dview = torch.as_strided(dbuf, (r, c), (ld, 1), off)
        return (view.T, dview.t()) if trans else (view, dview)
    A, dA = make(M, K, ta, offA, padA)
    B, dB = make(K, N, tb, offB, padB)
    ldc = N + padC
    bufC = draw(offC + M * ldc + 8)
    C0 = np.lib.stride_tricks.as_strided(bufC[offC:], (M, N), (ldc * bufC.itemsize, bufC.itemsize))
    dbufC = torch.from_numpy(bufC).cuda()
    dC = torch.as_strided(dbufC, (M, N), (ldc, 1), offC)
    # half of the float32 cases leave the choice to the library (cfg = -1): only then do the skinny, small-matrix,
    # slice-parallel and main + tail dispatches run at all
    cfg = int(rng.integers(0, ncfg)) if (dtype == np.float32 and rng.random() < 0.5) else -1
    mode = int(rng.random() < 0.3)
    laser_amd.set_f32_config(cfg); laser_amd.set_float_mode(mode)
    # launch plan of the assembly kernels (gemm_f32_asm.cpp): forced on, plain / persistent, random workgroup counts, both
    # receive paths, every tile, raster group -- laser-order results must not move by a bit under any of them
    plan = {}
    if not is_int and 0.45 <= shape_kind < 0.60:
        cfg = -1
        laser_amd.set_f32_config(-1)
        f64 = dtype == np.float64
        kerns = ([16, 18] if mode == 0 else [17, 19]) if f64 else ([0, 2, 30, 12, 46, 50, 54, 58, 62] if mode == 0 else [1, 8, 3, 31, 13, 47, 51, 55, 59, 63])      # (46..: the 16x16-block tiles; ineligible picks -- K % 4 != 0, transposed B -- fall to the model's own)
        plan = {"f64_asm" if f64 else "f32_asm": 2, "slice_parallel": 0, "asm_plan": int(rng.choice([0, 1, 2, 2, 3, 4])),
                "asm_kernel": int(rng.choice(kerns + [-1])), "asm_wgs": int(rng.choice([0, 0, 8 * int(rng.integers(1, 97))])),
                "asm_noseed": int(rng.random() < 0.4), "asm_group_m": int(rng.choice([0, 0, 1, 3, 8])),
                "asm_slice": int(rng.choice([0, 0, 4, 9])) if mode == 1 else 0}
        for k, v in plan.items():
            laser_amd.set_option(k, v)
    want = oracle.matmul(np.ascontiguousarray(A), np.ascontiguousarray(B), alpha=alpha, beta=beta, C_=np.ascontiguousarray(C0).copy(),
                         isa=oracle.fused_isa(dtype))
    laser_amd.matmul(dA, dB, alpha, beta, dC)
    got = dC.cpu().numpy()
    if plan:
        plan["used"] = (laser_amd.get_option("last_f64_asm" if dtype == np.float64 else "last_f32_asm"), laser_amd.get_option("last_asm_wgs"),
                        laser_amd.get_option("last_asm_slices"))
        for k, v in (("f32_asm", 1), ("f64_asm", 1), ("slice_parallel", 1), ("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("asm_noseed", 0),
                     ("asm_group_m", 0), ("asm_slice", 0)):
            laser_amd.set_option(k, v)
    untouched = np.array_equal(dbufC.cpu().numpy()[:offC], bufC[:offC])
    if mode == 0 or is_int:
        ok = np.array_equal(got, want)
    else:
        # FAST = one chain over K instead of Laser's kc slices: same products, different rounding points.  The mean
        # relative error of the reference's own check (error_functions.nim) is ill-conditioned when the results are
        # centred on zero, as they are here.  Norm-wise bound instead: |err_ij| <= c * eps * (|alpha| sum_k |a_ik||b_kj|
        # + |beta||c_ij|) -- the forward error of ANY summation order is gamma_K times that sum, in practice a few eps;
        # c = 8 keeps the FAST contract (1e-5 relative) honest: 8 eps = 9.5e-7 (f32)
        eps = np.finfo(dtype).eps
        Aa, Ba = np.abs(np.ascontiguousarray(A)).astype(np.float64), np.abs(np.ascontiguousarray(B)).astype(np.float64)
        bound = 8 * eps * (abs(float(alpha)) * (Aa @ Ba) + abs(float(beta)) * np.abs(np.ascontiguousarray(C0)).astype(np.float64)) + np.finfo(dtype).tiny
        ok = bool(np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound))
    if not (ok and untouched):
        fails += 1
        print("FAIL", dict(it=it, dtype=dtype.__name__, M=M, N=N, K=K, ta=ta, tb=tb, offs=(offA, offB, offC), pads=(padA, padB, padC),
                           alpha=float(alpha), beta=float(beta), cfg=cfg, mode=mode, plan=plan, nbad=int(np.sum(got != want))), flush=True)
laser_amd.set_f32_config(-1); laser_amd.set_float_mode(0)
print(f"fuzz: {cases} cases, {fails} failures, fix-up time-outs {laser_amd.get_option('asm_fixup_timeouts')}")
sys.exit(1 if fails else 0)
