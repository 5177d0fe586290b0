#!/usr/bin/env python3
"""Launch-plan sweep of the assembly kernels: for every shape and accumulation mode, every candidate tile
(option asm_kernel) under the plain plan (one tile per workgroup, asm_plan = 1) and the persistent plan (asm_plan = 2: every
workgroup slot an equal share of K-slice units, cut tiles finished by the in-kernel ordered fix-up), plus what the launcher's own
model picks (asm_plan = 0).  Timed through the C-ABI entry point bound once via ctypes.  One JSON line per shape and mode.
usage: plan_sweep.py [f32|f64] [shape-list name]"""
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
f64 = len(sys.argv) > 1 and sys.argv[1] == "f64"
which = sys.argv[2] if len(sys.argv) > 2 else "mid"
SHAPES = {
    "mid": [(1024,) * 3, (1280,) * 3, (1536,) * 3, (1792,) * 3, (1920,) * 3, (2048,) * 3, (2304,) * 3, (2560,) * 3, (3072,) * 3, (4096,) * 3,
            (1000, 3000, 2000), (1536, 1536, 4096), (4100,) * 3],
    "big": [(4096,) * 3, (6144,) * 3, (8192,) * 3],
    "f64": [(960,) * 3, (1024,) * 3, (1536,) * 3, (1792,) * 3, (2048,) * 3, (2304,) * 3, (4096,) * 3],
    # short-K problems with exactly three rounds of 256x128 tiles: the GEMM twins of C4's main launch (K = C_in * 9 = 1152)
    "shortk": [(8192, 3072, 1152), (8192, 3072, 576), (8192, 3072, 2304), (4096, 6144, 1152)],
    "small": [(512,) * 3, (640, 640, 4096), (768,) * 3, (896,) * 3, (1024, 1024, 8192), (512, 512, 8192)],
}[which]
PEAK = 78.6 if f64 else 157.3
dt = torch.float64 if f64 else torch.float32
fn = L.laser_hip_gemm_strided_f64_dev if f64 else L.laser_hip_gemm_strided_f32_dev
ct = ctypes.c_double if f64 else ctypes.c_float
CANDS = {0: ({16: "128x128", 18: "64x64"} if f64 else {0: "256x128", 2: "128x128", 30: "128x128x32", 12: "64x64"}),
         1: ({17: "128x128", 19: "64x64"} if f64 else {1: "256x256", 8: "256x128", 3: "128x128", 31: "128x128x32", 13: "64x64"})}
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
laser_amd.set_option("f64_asm" if f64 else "f32_asm", 2)


def timed(call, flops):
    call(); call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.025:          # clocks up
        for _ in range(4):
            call()
        torch.cuda.synchronize()
        n += 4
    inner = max(4, min(64, int(2e-3 / max(1e-6, flops / 100e12))))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[2]


for (M, N, K) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = ((torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2).to(dt)
    B = ((torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2).to(dt)
    C = torch.zeros((M, N), device="cuda", dtype=dt)
    cargs = (M, N, K, ct(1.0), ctypes.c_void_p(A.data_ptr()), K, 1, ctypes.c_void_p(B.data_ptr()), N, 1, ct(0.0), ctypes.c_void_p(C.data_ptr()), N, 1, stream)
    call = lambda: fn(*cargs)
    fl = 2.0 * M * N * K
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"dtype": "f64" if f64 else "f32", "M": M, "N": N, "K": K, "mode": "laser_order" if mode == 0 else "fast"}
        laser_amd.set_option("asm_kernel", -1)
        laser_amd.set_option("asm_plan", 0)
        ms = timed(call, fl)
        used = laser_amd.get_option("last_f64_asm") if f64 else laser_amd.last_f32_asm()
        rec["auto"] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / PEAK, 3), "kernel": used,
                       "wgs": laser_amd.get_option("last_asm_wgs"), "slices": laser_amd.get_option("last_asm_slices")}
        ref = C.clone() if mode == 0 else None
        for kern, name in CANDS[mode].items():
            laser_amd.set_option("asm_kernel", kern)
            for plan in (1, 2):
                laser_amd.set_option("asm_plan", plan)
                ms = timed(call, fl)
                used = laser_amd.get_option("last_f64_asm") if f64 else laser_amd.last_f32_asm()
                if used != kern + 1:
                    continue
                ent = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "wgs": laser_amd.get_option("last_asm_wgs"), "slices": laser_amd.get_option("last_asm_slices")}
                if mode == 0:
                    ent["same_bits"] = bool(torch.equal(ref, C))
                rec[f"{name}/{'plain' if plan == 1 else 'persistent'}"] = ent
        cands = [(v["tflops"], k) for k, v in rec.items() if isinstance(v, dict) and k != "auto"]
        if cands:
            best = max(cands)
            rec["best"] = best[1]
            rec["auto_vs_best"] = round(rec["auto"]["tflops"] / best[0], 3)
        print(json.dumps(rec), flush=True)
laser_amd.set_option("asm_kernel", -1); laser_amd.set_option("asm_plan", 0); laser_amd.set_float_mode(0)
laser_amd.set_option("f64_asm" if f64 else "f32_asm", 1)
print(json.dumps({"fixup_timeouts": laser_amd.get_option("asm_fixup_timeouts")}))
