#!/usr/bin/env python3
"""Randomised cases through the interpreter (laser_amd/asmgen/sim.py) for every shipped assembly kernel -- f32 GEMM (all tiles, plain /
transposed B, alpha / beta, batches, fused bias / relu), f64 (alpha / beta, batches), int32 / int64 (alpha / beta), 3x3 convolutions (any
padding, bias / relu): addresses, layouts, counted waits and hazards of the generated programs, no GPU needed.
usage: sim_fuzz.py [seed] [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laser_amd.asmgen import check as C, f32_kernel as K
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600
gemm = [n for n in K.CONFIGS if not n.startswith("conv")]
t0 = time.time(); n = fails = 0
while time.time() - t0 < T:
    kind = rng.choice(["f32", "f32", "f64", "i32", "i64", "conv"])
    try:
        if kind == "f32":
            name = str(rng.choice(gemm)); c = K.CONFIGS[name]
            M, N = int(rng.integers(1, 2 * c["BM"] + 20)), int(rng.integers(1, 2 * c["BN"] + 20))
            Kd = int(rng.choice([rng.integers(1, 70), rng.integers(500, 1100)]))
            kw = dict(lda=Kd + int(rng.integers(0, 5)), ldc=N + int(rng.integers(0, 5)), alpha=float(rng.choice([1.0, 0.5, -2.0])), beta=float(rng.choice([0.0, 0.0, 1.0, 0.25])),
                      batch=int(rng.choice([1, 1, 2])), seed=int(rng.integers(1 << 30)))
            if c.get("b_kcontig"): kw["ldb"] = Kd + int(rng.integers(0, 5))
            else: kw["ldb"] = N + int(rng.integers(0, 5))
            if rng.random() < 0.3 and kw["beta"] == 0.0 or c.get("exact", False):
                if rng.random() < 0.3: kw.update(bias=str(rng.choice(["row", "col", "full"])), act=int(rng.integers(0, 2)))
            ok = C.run_case(name, M, N, Kd, verbose=False, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "f64":
            from laser_amd.asmgen import f64_kernel as K64
            name = str(rng.choice(list(K64.CONFIGS))); c = K64.CONFIGS[name]
            M, N = int(rng.integers(1, 2 * c["BM"] + 10)), int(rng.integers(1, 2 * c["BN"] + 10))
            Kd = 2 * int(rng.choice([rng.integers(1, 40), rng.integers(130, 300)]))
            kw = dict(lda=Kd + int(rng.integers(0, 4)), ldc=N + int(rng.integers(0, 4)), alpha=float(rng.choice([1.0, 0.5, -2.0])), beta=float(rng.choice([0.0, 1.0, 0.25])),
                      batch=int(rng.choice([1, 2])), seed=int(rng.integers(1 << 30)))
            kw["ldb"] = (Kd if c.get("b_kcontig") else N) + int(rng.integers(0, 4))
            ok = C.run_case64(name, M, N, Kd, verbose=False, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "i32":
            M, N, Kd = int(rng.integers(1, 270)), int(rng.integers(1, 270)), int(rng.integers(1, 200))
            kw = dict(ldc=N + int(rng.integers(0, 4)), alpha=int(rng.choice([1, -3, 2**31 - 1])), beta=int(rng.choice([0, 1, 7])), seed=int(rng.integers(1 << 30)))
            ok = C.run_case_i32(M, N, Kd, verbose=False, **kw); desc = ("i32", M, N, Kd, kw)
        elif kind == "i64":
            M, N, Kd = int(rng.integers(1, 140)), int(rng.integers(1, 140)), int(rng.integers(1, 150))
            kw = dict(ldc=N + int(rng.integers(0, 4)), alpha=int(rng.choice([1, -3, 2**63 - 1])), beta=int(rng.choice([0, 1, -7])), seed=int(rng.integers(1 << 30)))
            ok = C.run_case_i64(M, N, Kd, verbose=False, **kw); desc = ("i64", M, N, Kd, kw)
        else:
            name = str(rng.choice([n for n in K.CONFIGS if n.startswith("conv")])); c = K.CONFIGS[name]
            Cin = 4 * int(rng.integers(1, 5)) if rng.random() < 0.7 else 4 * int(rng.integers(14, 18))
            H, W = int(rng.integers(3, 14)), 2 * int(rng.integers(2, 9))
            pad = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
            M = int(rng.integers(1, c["BM"] + 30))
            kw = dict(seed=int(rng.integers(1 << 30)), bias=bool(rng.random() < 0.3), act=int(rng.integers(0, 2)))
            ok = C.run_conv_case(name, int(rng.integers(1, 3)), Cin, H, W, M, pad, verbose=False, **kw); desc = (name, Cin, H, W, M, pad, kw)
    except AssertionError as e:
        ok = False; desc = ("ASSERT", kind, str(e)[:200])
    n += 1
    if not ok:
        fails += 1; print("FAIL", desc, flush=True)
print(f"sim fuzz: {n} cases, {fails} failures, {time.time() - t0:.0f} s")
