#!/usr/bin/env python3
"""Randomised cases through the interpreter (laser_amd/asmgen/sim.py) for every shipped assembly kernel -- f32 GEMM (all tiles, plain /
transposed B, alpha / beta, batches, fused bias / relu), f64 (alpha / beta, batches), int32 / int64 (alpha / beta), 3x3 convolutions (any
padding, bias / relu): addresses, layouts, counted waits and hazards of the generated programs, no GPU needed.
Round 4: + persistent launches with K-slice cuts (one- and two-level ranges, both receive paths, XCD remap, raster groups; f32 and f64),
strided C views, the fused-prologue kernels, the two-tile prefetch option.
Round 6: + the 16x16-block tile family, the convolution unit walkers, the hybrid two-launch plan, the float64 strided plan.
usage: sim_fuzz.py [seed] [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laser_amd.asmgen import check as C, f32_kernel as K
C.BANK_MODEL = False      # (bank-conflict statistics: a quarter of the interpreter's time, not a correctness check)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600
gemm = [n for n in K.CONFIGS if not n.startswith("conv")]
t0 = time.time(); n = fails = 0
while time.time() - t0 < T:
    kind = rng.choice(["f32", "f32", "f64", "i32", "i64", "conv", "sched", "sched", "sched64", "view", "x16", "x16", "walk", "walk", "hybrid", "hybrid", "pipe64"])
    if os.environ.get("SIM_FUZZ_VERBOSE"):
        print("start", kind, flush=True)
    try:
        if kind == "walk":
            # round 6: the convolution kernels as unit walkers (Cfg.cpers): G workgroups over images x tiles units; K a multiple of 32
            # (pipelined transitions) or not (every unit through the epilogue), any kernel / stride / padding
            name = str(rng.choice([n_ for n_ in K.CONFIGS if n_.startswith("conv") and n_.endswith("_p")])); c = K.CONFIGS[name]
            kH, kW = (1, 1) if rng.random() < 0.3 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
            Cin = int(rng.choice([32, 64, 96])) if rng.random() < 0.7 else 4 * int(rng.integers(1, 20))
            if rng.random() < 0.3: Cin = 32 * int(rng.integers(1, 3)); kH = kW = 3
            H, W = int(rng.integers(max(3, kH), 18)), int(rng.integers(max(3, kW), 22))
            pad = (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
            stride = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
            images = int(rng.integers(1, 4))
            M = int(rng.integers(1, c["BM"] + 40))
            oH, oW = (H + 2 * pad[0] - kH) // stride[0] + 1, (W + 2 * pad[1] - kW) // stride[1] + 1
            units = images * (-(-M // c["BM"])) * (-(-(oH * oW) // 128))
            kw = dict(seed=int(rng.integers(1 << 30)), kernel=(kH, kW), stride=stride, G=int(rng.integers(1, units + 1)), bias=bool(rng.random() < 0.15), act=int(rng.random() < 0.15))
            ok = C.run_conv_case(name, images, Cin, H, W, M, pad, verbose=False, **kw); desc = (name, images, Cin, H, W, M, pad, kw)
        elif kind == "hybrid":
            # round 6: the two-launch plan -- whole rounds strided, the remaining tiles cut along K with the kernels' tile base (f32, 16x16-block
            # tiles): any K (transitions are only taken for whole K-tiles), both receive paths, raster groups
            from laser_amd.asmgen import f32x16_kernel as K16
            mod = K16 if rng.random() < 0.3 else K
            name = str(rng.choice([n_ for n_ in mod.CONFIGS if not n_.startswith("conv") and "_pre" not in n_ and "256x256" not in n_])); c = mod.CONFIGS[name]
            exact = bool(c.get("exact", False))
            tm, tn = int(rng.integers(1, 4)), int(rng.integers(2, 5))
            Tt = tm * tn
            cand = [g_ for g_ in range(2, Tt) if Tt % g_]
            if not cand:
                tm, tn = 1, 3; Tt = 3; cand = [2]
            G = int(rng.choice(cand))
            M, N = (tm - 1) * c["BM"] + int(rng.integers(1, c["BM"] + 1)), (tn - 1) * c["BN"] + int(rng.integers(1, c["BN"] + 1))
            Kd = 4 * int(rng.choice([rng.integers(130, 270), rng.integers(24, 60), 8 * int(rng.integers(3, 20))]))
            P = -(-Kd // 512) if exact else max(1, -(-Kd // (32 * 5)))
            R = Tt % G
            kw = dict(G=G, hybrid=int(rng.integers(1, R * P + 1)), split=(True if exact else 5), seed=int(rng.integers(1 << 30)), noseed=int(rng.random() < 0.4))
            if not exact: kw.update(integer=True)
            if rng.random() < 0.5: kw["group_m"] = int(rng.integers(1, tm + 1))
            if mod is K16: kw["mod"] = K16
            ok = C.run_case(name, M, N, Kd, verbose=False, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "pipe64":
            # round 6: the float64 kernels' strided plan (pipelined transitions when K is a multiple of 16, three K-tiles or more, beta == 0)
            from laser_amd.asmgen import f64_kernel as K64
            name = str(rng.choice(list(K64.CONFIGS))); c = K64.CONFIGS[name]
            tm, tn = int(rng.integers(1, 4)), int(rng.integers(1, 4))
            M, N = (tm - 1) * c["BM"] + int(rng.integers(1, c["BM"] + 1)), (tn - 1) * c["BN"] + int(rng.integers(1, c["BN"] + 1))
            Kd = int(rng.choice([16 * int(rng.integers(3, 20)), 2 * int(rng.integers(20, 160))]))
            kw = dict(G=int(rng.integers(1, tm * tn + 1)), strided=True, seed=int(rng.integers(1 << 30)), alpha=float(rng.choice([1.0, 1.0, 0.5])), beta=float(rng.choice([0.0, 0.0, 0.0, 2.0])))
            if rng.random() < 0.4: kw["group_m"] = int(rng.integers(1, tm + 1))
            ok = C.run_case64(name, M, N, Kd, verbose=False, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "x16":
            # round 6: the 16x16-block tile family (f32x16_kernel.py): plain and persistent K-cut launches, K a multiple of 4
            from laser_amd.asmgen import f32x16_kernel as K16
            name = str(rng.choice(list(K16.CONFIGS))); c = K16.CONFIGS[name]
            exact = bool(c.get("exact", False))
            if rng.random() < 0.5:
                M, N = int(rng.integers(1, 2 * c["BM"] + 20)), int(rng.integers(1, 2 * c["BN"] + 20))
                Kd = 4 * int(rng.choice([rng.integers(1, 18), rng.integers(125, 275)]))
                kw = dict(lda=Kd + int(rng.integers(0, 5)), ldc=N + int(rng.integers(0, 5)), alpha=float(rng.choice([1.0, 0.5, -2.0])), beta=float(rng.choice([0.0, 0.0, 1.0, 0.25])),
                          batch=int(rng.choice([1, 1, 2])), seed=int(rng.integers(1 << 30)))
                kw["ldb"] = (Kd if c.get("b_kcontig") else N) + int(rng.integers(0, 5))
            else:
                tm, tn = int(rng.integers(1, 4)), int(rng.integers(1, 4))
                M, N = (tm - 1) * c["BM"] + int(rng.integers(1, c["BM"] + 1)), (tn - 1) * c["BN"] + int(rng.integers(1, c["BN"] + 1))
                Kd = 4 * int(rng.choice([rng.integers(129, 270), rng.integers(257, 530), rng.integers(1, 128)]))
                P = -(-Kd // 512) if exact else max(1, -(-Kd // (32 * 5)))
                units = tm * tn * P
                kw = dict(G=int(rng.integers(1, units + 1)), split=(True if exact else 5), seed=int(rng.integers(1 << 30)), noseed=int(rng.random() < 0.4),
                          alpha=float(rng.choice([1.0, 1.0, 0.5])), beta=float(rng.choice([0.0, 0.0, 0.25])))
                if not exact: kw.update(alpha=1.0, beta=float(rng.choice([0.0, 1.0])), integer=True)
                if rng.random() < 0.5: kw["group_m"] = int(rng.integers(1, tm + 1))
                if kw["G"] >= 8 and rng.random() < 0.5: kw["xcd"] = True
            ok = C.run_case(name, M, N, Kd, verbose=False, mod=K16, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "f32":
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
        elif kind in ("sched", "sched64"):
            # persistent launches: G workgroups share tiles x K-slice units; cut tiles handed over in-kernel (one-level and two-level
            # ranges, either receive path, XCD remap, raster groups); one-chain kernels on integer-valued operands (their cut moves the
            # rounding points, not the sum); a fifth of the f32 cases on the two-tile prefetch option
            f64 = kind == "sched64"
            if f64:
                from laser_amd.asmgen import f64_kernel as K64
                name = str(rng.choice([n_ for n_ in K64.CONFIGS])); c = K64.CONFIGS[name]; kc, bk = 256, c["BK"]      # (its operands are integer-valued: every cut is exact)
            else:
                name = str(rng.choice([n_ for n_ in gemm if "_pre" not in n_])); c = K.CONFIGS[name]; kc, bk = 512, c["BK"]
            tm, tn = int(rng.integers(1, 4)), int(rng.integers(1, 4))
            if rng.random() < 0.25: tm, tn = int(rng.integers(2, 5)), int(rng.integers(2, 5))
            M, N = (tm - 1) * c["BM"] + int(rng.integers(1, c["BM"] + 1)), (tn - 1) * c["BN"] + int(rng.integers(1, c["BN"] + 1))
            Kd = int(rng.choice([rng.integers(kc + 1, 2 * kc + 40), rng.integers(2 * kc + 1, 4 * kc + 60), rng.integers(1, kc)]))
            if f64: Kd += Kd & 1
            exact = bool(c.get("exact", False))
            P = -(-Kd // kc) if exact else max(1, -(-Kd // (bk * 5)))
            units = tm * tn * P
            G = int(rng.integers(1, units + 1))
            kw = dict(G=G, split=(True if exact else 5), seed=int(rng.integers(1 << 30)), noseed=int(rng.random() < 0.4),
                      alpha=float(rng.choice([1.0, 1.0, 0.5])), beta=float(rng.choice([0.0, 0.0, 0.25])))
            if not exact: kw.update(alpha=1.0, beta=float(rng.choice([0.0, 1.0])))
            if not exact and not f64: kw["integer"] = True
            if rng.random() < 0.5: kw["group_m"] = int(rng.integers(1, tm + 1))
            if tm * tn >= 8 and rng.random() < 0.5:
                G = 8 * int(rng.integers(1, max(2, min(units, 4 * tm * tn) // 8 + 1)))
                if (tm * tn // 8) * P >= G // 8: kw.update(G=G, two_level=True, xcd=True)
            elif kw["G"] >= 8 and rng.random() < 0.5:
                kw["xcd"] = True
            if not f64 and rng.random() < 0.2 and c["BM"] == 64: kw["over"] = {"deep": True}
            ok = (C.run_case64 if f64 else C.run_case)(name, M, N, Kd, verbose=False, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "view":
            # C views (column stride, interleaved rows) and the fused prologue's kernel variants
            if rng.random() < 0.5:
                name = str(rng.choice([n_ for n_ in gemm if "_pre" not in n_])); c = K.CONFIGS[name]
                M, N = int(rng.integers(1, 2 * c["BM"] + 20)), int(rng.integers(1, 2 * c["BN"] + 20))
                Kd = int(rng.choice([rng.integers(1, 70), rng.integers(500, 1100)]))
                csc = int(rng.integers(2, 4))
                kw = dict(csc=csc, seed=int(rng.integers(1 << 30)), beta=float(rng.choice([0.0, 0.5])))
                # (rows must be the slow direction of the view the kernel sees: rowStride >= (N - 1) * colStride + 1 -- the launcher
                # runs the transposed product otherwise, gemm_f32_asm.cpp)
                kw.update(ldc=(N - 1) * csc + 1 + int(rng.integers(0, 4)))
            else:
                name = str(rng.choice([n_ for n_ in gemm if "_pre" in n_])); c = K.CONFIGS[name]
                M, N = int(rng.integers(1, 2 * c["BM"] + 20)), int(rng.integers(1, 2 * c["BN"] + 20))
                Kd = int(rng.choice([rng.integers(1, 70), rng.integers(500, 1100)]))
                kw = dict(pre=int(rng.integers(1, 4)), seed=int(rng.integers(1 << 30)), beta=float(rng.choice([0.0, 0.25])))
            ok = C.run_case(name, M, N, Kd, verbose=False, **kw); desc = (name, M, N, Kd, kw)
        elif kind == "i32":
            M, N, Kd = int(rng.integers(1, 270)), int(rng.integers(1, 270)), int(rng.integers(1, 200))
            kw = dict(ldc=N + int(rng.integers(0, 4)), alpha=int(rng.choice([1, -3, 2**31 - 1])), beta=int(rng.choice([0, 1, 7])), seed=int(rng.integers(1 << 30)))
            ok = C.run_case_i32(M, N, Kd, verbose=False, **kw); desc = ("i32", M, N, Kd, kw)
        elif kind == "i64":
            M, N, Kd = int(rng.integers(1, 140)), int(rng.integers(1, 140)), int(rng.integers(1, 150))
            kw = dict(ldc=N + int(rng.integers(0, 4)), alpha=int(rng.choice([1, -3, 2**63 - 1])), beta=int(rng.choice([0, 1, -7])), seed=int(rng.integers(1 << 30)))
            ok = C.run_case_i64(M, N, Kd, verbose=False, **kw); desc = ("i64", M, N, Kd, kw)
        else:
            name = str(rng.choice([n for n in K.CONFIGS if n.startswith("conv") and not n.endswith("_p")])); c = K.CONFIGS[name]
            Cin = 4 * int(rng.integers(1, 5)) if rng.random() < 0.7 else 4 * int(rng.integers(14, 18))
            H, W = int(rng.integers(3, 14)), 2 * int(rng.integers(2, 9))
            pad = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
            M = int(rng.integers(1, c["BM"] + 30))
            kw = dict(seed=int(rng.integers(1 << 30)), bias=bool(rng.random() < 0.3), act=int(rng.integers(0, 2)))
            ok = C.run_conv_case(name, int(rng.integers(1, 3)), Cin, H, W, M, pad, verbose=False, **kw); desc = (name, Cin, H, W, M, pad, kw)
    except (AssertionError, TypeError, KeyError, ValueError) as e:
        ok = False; desc = (type(e).__name__, kind, str(e)[:200])
    n += 1
    if os.environ.get("SIM_FUZZ_VERBOSE"):
        print(n, kind, desc, "OK" if ok else "FAIL", flush=True)
    if not ok:
        fails += 1; print("FAIL", desc, flush=True)
print(f"sim fuzz: {n} cases, {fails} failures, {time.time() - t0:.0f} s")
