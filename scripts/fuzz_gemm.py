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
    cfg = int(rng.integers(-1, ncfg)) if dtype == np.float32 else -1
    mode = int(rng.random() < 0.3)
    laser_amd.set_f32_config(cfg); laser_amd.set_float_mode(mode)
    want = oracle.matmul(np.ascontiguousarray(A), np.ascontiguousarray(B), alpha=alpha, beta=beta, C_=np.ascontiguousarray(C0).copy(),
                         isa=oracle.fused_isa(dtype))
    laser_amd.matmul(dA, dB, alpha, beta, dC)
    got = dC.cpu().numpy()
    untouched = np.array_equal(dbufC.cpu().numpy()[:offC], bufC[:offC])
    if mode == 0 or is_int:
        ok = np.array_equal(got, want)
    else:
        # FAST = one chain over K instead of Laser's kc slices: same products, different rounding points.  The mean
        # relative error of the reference's own check (error_functions.nim) is ill-conditioned when the results are
        # centred on zero, as they are here; bound the absolute deviation by the size of the rounding noise instead
        ok = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64)))) <= 4e-7 * K * (1.0 + abs(float(beta)))
    if not (ok and untouched):
        fails += 1
        print("FAIL", dict(it=it, dtype=dtype.__name__, M=M, N=N, K=K, ta=ta, tb=tb, offs=(offA, offB, offC), pads=(padA, padB, padC),
                           alpha=float(alpha), beta=float(beta), cfg=cfg, mode=mode, nbad=int(np.sum(got != want))), flush=True)
laser_amd.set_f32_config(-1); laser_amd.set_float_mode(0)
print(f"fuzz: {cases} cases, {fails} failures")
sys.exit(1 if fails else 0)
