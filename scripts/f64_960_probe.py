#!/usr/bin/env python3
"""f64 960^3 (the reference's float64 bench shape, gemm_bench_float64.nim:252-258): the plain launch (225 tiles of 64x64, one
workgroup on 225 of 256 CUs) against K-cut plans with forced workgroup counts -- two workgroups per CU cover each other's waits, the
sender / receiver pairs fold their kc slices in order (same bits).  One JSON line per variant.  usage: f64_960_probe.py [n] [reps]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from laser_amd import _lib as _lh

L = _lh.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.rand((n, n), generator=g, device="cuda", dtype=torch.float64) - 0.5
B = torch.rand((n, n), generator=g, device="cuda", dtype=torch.float64) - 0.5
C = torch.zeros((n, n), device="cuda", dtype=torch.float64)
fn = L.laser_hip_gemm_strided_f64_dev
ct = ctypes.c_double
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
cargs = (n, n, n, ct(1.0), ctypes.c_void_p(A.data_ptr()), n, 1, ctypes.c_void_p(B.data_ptr()), n, 1, ct(0.0), ctypes.c_void_p(C.data_ptr()), n, 1, stream)
call = lambda: fn(*cargs)
fl = 2.0 * n ** 3


def timed(inner=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


variants = [("model", 0, 0), ("plain", 1, 0)] + [(f"cut G={w}", 2, w) for w in (256, 384, 448, 456, 512)]
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    ref, recs, times = None, {}, {}
    for name, plan, wgs in variants:
        laser_amd.set_option("asm_plan", plan); laser_amd.set_option("asm_wgs", wgs)
        C.fill_(float("nan"))
        rc = call(); torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        recs[name] = {"rc": rc, "wgs": laser_amd.get_option("last_asm_wgs"),
                      "slices": laser_amd.get_option("last_asm_slices"), "same_bits_as_model": bool(torch.equal(C, ref))}
        times[name] = []
    for _ in range(100):
        call()
    for _ in range(reps):
        for name, plan, wgs in variants:
            laser_amd.set_option("asm_plan", plan); laser_amd.set_option("asm_wgs", wgs)
            call()
            times[name].append(timed())
    for name in recs:
        t = sorted(times[name]); ms = t[len(t) // 2]
        recs[name].update({"us": round(ms * 1e3, 2), "tflops": round(fl / ms / 1e9, 2)})
    vend = []
    for _ in range(reps):
        torch.matmul(A, B, out=C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            torch.matmul(A, B, out=C)
        e1.record(); torch.cuda.synchronize(); vend.append(e0.elapsed_time(e1) / 40)
    vend.sort()
    print(json.dumps({"n": n, "mode": "laser_order" if mode == 0 else "fast", "variants": recs, "vendor_us": round(vend[len(vend) // 2] * 1e3, 2),
                      "vendor_tflops": round(fl / vend[len(vend) // 2] / 1e9, 2)}), flush=True)
laser_amd.set_option("asm_plan", 0); laser_amd.set_option("asm_wgs", 0); laser_amd.set_float_mode(0)
