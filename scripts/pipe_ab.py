#!/usr/bin/env python3
"""Round 6: pipelined tile transitions (asmgen/f32_kernel.py Cfg.pipe) A/B.  For every shape, accumulation mode and forced tile kernel:
the plain launch (asm_plan = 1: one workgroup per tile) against the strided persistent launch (asm_plan = 3: workgroup v walks tiles
v, v + G, ... without leaving its K loop), interleaved and repeated; the two results must be the same bits.  One JSON line per
(shape, mode, kernel).  usage: pipe_ab.py [shape-list name] [reps]"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from laser_amd import _lib as _lh

L = _lh.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "big"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SHAPES = {
    "big": [(8192,) * 3, (4096,) * 3, (8192, 3072, 1152), (6144,) * 3],
    "head": [(8192,) * 3],
    "shortk": [(8192, 3072, 1152), (8192, 3072, 576), (8192, 3072, 2304), (4096, 6144, 1152)],
    "mid": [(2560,) * 3, (3072,) * 3, (4096,) * 3, (4100, 4100, 4096), (5120,) * 3],
    "frac": [(6912,) * 3, (6656,) * 3, (5888,) * 3, (2816,) * 3, (4608,) * 3, (7424,) * 3],      # tile counts a fraction of a round above whole rounds
    "x16": [(3072,) * 3, (4608,) * 3, (5120,) * 3, (7680,) * 3, (8192, 3072, 1152)],
}[which]
CANDS = {0: {0: "256x128", 30: "128x128x32", 2: "128x128", 12: "64x64"}, 1: {1: "256x256", 8: "256x128", 31: "128x128x32", 3: "128x128"}}
if which == "x16":      # the 16x16-block tiles (f32x16_kernel.py), pipelined since round 6
    CANDS = {0: {46: "96x96 (x16)", 50: "160x96 (x16)", 54: "128x96 (x16)", 58: "192x96 (x16)", 62: "160x160 (x16)"},
             1: {47: "96x96 (x16)", 51: "160x96 (x16)", 55: "128x96 (x16)", 59: "192x96 (x16)", 63: "160x160 (x16)"}}
if len(sys.argv) > 3:      # only these kernel indices
    keep = {int(x) for x in sys.argv[3].split(",")}
    CANDS = {m: {k: v for k, v in d.items() if k in keep} for m, d in CANDS.items()}
fn = L.laser_hip_gemm_strided_f32_dev
ct = ctypes.c_float
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
laser_amd.set_option("f32_asm", 2)


def warm(call):
    call(); call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        for _ in range(4):
            call()
        torch.cuda.synchronize()


def timed(call, flops):
    inner = max(4, min(64, int(3e-3 / max(1e-6, flops / 100e12))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


for (M, N, K) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2
    B = (torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    cargs = (M, N, K, ct(1.0), ctypes.c_void_p(A.data_ptr()), K, 1, ctypes.c_void_p(B.data_ptr()), N, 1, ct(0.0), ctypes.c_void_p(C.data_ptr()), N, 1, stream)
    call = lambda: fn(*cargs)
    fl = 2.0 * M * N * K
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        for kern, kname in CANDS[mode].items():
            laser_amd.set_option("asm_kernel", kern)
            rec = {"M": M, "N": N, "K": K, "mode": "laser_order" if mode == 0 else "fast", "kernel": kname}
            PLANS = (1, 3, 2) if "cut" in sys.argv[4:] else (1, 3)      # 2 = the persistent plan with K-slice cuts and hand-overs
            if "hyb" in sys.argv[4:]:
                PLANS = PLANS + (4,)                                      # 4 = strided whole rounds + a K-cut launch over the remaining tiles
            outs, times, meta = {}, {p_: [] for p_ in PLANS}, {}
            ok = True
            for plan in PLANS:
                laser_amd.set_option("asm_plan", plan)
                C.fill_(float("nan"))
                rc = call()
                torch.cuda.synchronize()
                if rc != 0 or laser_amd.last_f32_asm() != kern + 1:
                    ok = False
                    break
                outs[plan] = C.clone()
                meta[plan] = {"wgs": laser_amd.get_option("last_asm_wgs"), "slices": laser_amd.get_option("last_asm_slices")}
                if plan == 4:
                    meta[plan]["rem_tiles"] = laser_amd.get_option("last_asm_rem")
            if not ok:
                rec["skipped"] = "kernel not eligible"
                print(json.dumps(rec), flush=True)
                continue
            rec["bit_identical"] = bool(torch.equal(outs[1], outs[3])) and all(q_ not in outs or mode == 1 or bool(torch.equal(outs[1], outs[q_])) for q_ in (2, 4))
            rec["finite"] = bool(torch.isfinite(outs[3]).all())
            laser_amd.set_option("asm_plan", 1)
            warm(call)
            for _ in range(reps):
                for plan in PLANS:
                    laser_amd.set_option("asm_plan", plan)
                    call()
                    times[plan].append(timed(call, fl))
            for plan, nm in [(p_, {1: "plain", 3: "pipe", 2: "cut", 4: "hybrid"}[p_]) for p_ in PLANS]:
                ts = sorted(times[plan])
                ms = ts[len(ts) // 2]
                rec[nm] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / 157.3, 4), "min_ms": round(ts[0], 4), **meta[plan]}
            rec["gain_pct"] = round(100.0 * (rec["plain"]["ms"] / rec["pipe"]["ms"] - 1.0), 2)
            print(json.dumps(rec), flush=True)
laser_amd.set_option("asm_kernel", -1)
laser_amd.set_option("asm_plan", 0)
