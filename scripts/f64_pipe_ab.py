#!/usr/bin/env python3
"""Round 6: pipelined tile transitions in the float64 kernels (f64_kernel.py trans_after): the plain launch (asm_plan = 1: a workgroup
per tile) against the strided persistent launch (asm_plan = 3: workgroup v walks tiles v, v + G, ... without leaving its K loop) and
the library's own choice, interleaved; same bits required (laser-order: against each other; both modes: plain == strided).
One JSON line per (size, mode).  usage: f64_pipe_ab.py [sizes, comma-separated] [reps]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from laser_amd import _lib as _lh

L = _lh.lib()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 3072, 4096, 6144, 8192]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fn = L.laser_hip_gemm_strided_f64_dev
ct = ctypes.c_double
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for n in sizes:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand((n, n), generator=g, device="cuda", dtype=torch.float64) - 0.5
    B = torch.rand((n, n), generator=g, device="cuda", dtype=torch.float64) - 0.5
    C = torch.zeros((n, n), device="cuda", dtype=torch.float64)
    cargs = (n, n, n, ct(1.0), ctypes.c_void_p(A.data_ptr()), n, 1, ctypes.c_void_p(B.data_ptr()), n, 1, ct(0.0), ctypes.c_void_p(C.data_ptr()), n, 1, stream)
    call = lambda: fn(*cargs)
    fl = 2.0 * n ** 3
    inner = max(2, min(40, int(4e-3 / (fl / 60e12))))

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / inner

    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"n": n, "mode": "laser_order" if mode == 0 else "fast"}
        outs, ts = {}, {}
        variants = (("plain", 1), ("strided", 3), ("model", 0))
        for name, plan in variants:
            laser_amd.set_option("asm_plan", plan)
            C.fill_(float("nan"))
            rc = call(); torch.cuda.synchronize()
            outs[name] = C.clone()
            rec[name] = {"rc": rc, "kernel": laser_amd.get_option("last_f64_asm"), "wgs": laser_amd.get_option("last_asm_wgs"), "slices": laser_amd.get_option("last_asm_slices")}
            ts[name] = []
        for _ in range(10):
            call()
        for _ in range(reps):
            for name, plan in variants:
                laser_amd.set_option("asm_plan", plan)
                call()
                ts[name].append(timed())
        for name, _ in variants:
            t = sorted(ts[name]); ms = t[len(t) // 2]
            rec[name].update({"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2), "frac": round(fl / ms / 1e9 / 78.6, 4)})
        rec["plain_eq_strided"] = bool(torch.equal(outs["plain"], outs["strided"]))
        rec["model_eq_plain"] = bool(torch.equal(outs["plain"], outs["model"]))
        rec["gain_pct"] = round(100.0 * (rec["plain"]["ms"] / rec["strided"]["ms"] - 1.0), 2)
        print(json.dumps(rec), flush=True)
laser_amd.set_option("asm_plan", 0)
laser_amd.set_float_mode(0)
