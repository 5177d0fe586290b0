#!/usr/bin/env python3
"""Round 6: the 16x16-block tile family (asmgen/f32x16_kernel.py: 96x96, 160x96) against the 32x32-block tiles.  For every shape and
accumulation mode: each candidate kernel forced (option asm_kernel, plain one-tile-per-workgroup launches and whatever plan the model
takes), timed interleaved; laser-order results must be the SAME BITS whatever the tile (every element is the same kc-sliced fmaf
chain, gemm.nim:150-158) -- the 32x32-block kernels are the ones the parity suite pins against the oracle.  Last per shape: what the
launch model picks on its own.  One JSON line per (shape, mode).  usage: x16_ab.py [shape-list] [reps] [nt]"""
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
which = sys.argv[1] if len(sys.argv) > 1 else "mid"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nt = len(sys.argv) > 3 and sys.argv[3] == "nt"
SHAPES = {
    "mid": [(1536,) * 3, (1920,) * 3, (2048,) * 3, (2560,) * 3, (3072,) * 3, (1000, 3000, 2000), (4100, 4100, 4100), (5120,) * 3],
    "ref": [(1920,) * 3, (1536,) * 3],
    "more": [(1664,) * 3, (2304,) * 3, (2560,) * 3, (2944,) * 3, (3072,) * 3, (3584,) * 3, (5120,) * 3, (1000, 3000, 2000), (4100, 4100, 4100)],
    "small": [(768,) * 3, (960,) * 3, (1152,) * 3, (1344,) * 3],
    "big2": [(5632,) * 3, (6912,) * 3, (7936,) * 3, (3328,) * 3],
    "big": [(4096,) * 3, (6144,) * 3, (8192,) * 3],
}[which]
# mode -> kernel index -> name (gemm_f32_asm.cpp kKernels)
o = 4 if nt else 0
o2 = 2 if nt else 0
CANDS = {0: {0 + o: "256x128x32", 2 + o: "128x128x16", 30 + o2: "128x128x32", 12 + o2: "64x64x32", 46 + o2: "96x96x32 (16x16 blocks)", 50 + o2: "160x96x32 (16x16 blocks)",
             54 + o2: "128x96x32 (16x16 blocks)", 58 + o2: "192x96x32 (16x16 blocks)", 62 + o2: "160x160x32 (16x16 blocks)"},
         1: {1 + o: "256x256x16", (9 if nt else 8): "256x128x32", 3 + o: "128x128x16", 31 + o2: "128x128x32", 13 + o2: "64x64x32", 47 + o2: "96x96x32 (16x16 blocks)",
             51 + o2: "160x96x32 (16x16 blocks)", 55 + o2: "128x96x32 (16x16 blocks)", 59 + o2: "192x96x32 (16x16 blocks)", 63 + o2: "160x160x32 (16x16 blocks)"}}
fn = L.laser_hip_gemm_strided_f32_dev
ct = ctypes.c_float
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def warm(call):
    call(); call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
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
    B = (torch.rand((N, K) if nt else (K, N), generator=g, device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    rsB, csB = (1, K) if nt else (N, 1)
    cargs = (M, N, K, ct(1.0), ctypes.c_void_p(A.data_ptr()), K, 1, ctypes.c_void_p(B.data_ptr()), rsB, csB, ct(0.0), ctypes.c_void_p(C.data_ptr()), N, 1, stream)
    call = lambda: fn(*cargs)
    fl = 2.0 * M * N * K
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"M": M, "N": N, "K": K, "B": "transposed" if nt else "row-major", "mode": "laser_order" if mode == 0 else "fast", "kernels": {}}
        ref = None
        runs = []       # (label, setup)
        for kern, kname in CANDS[mode].items():
            for plan in (1, 0):
                runs.append((f"{kname} / {'plain' if plan == 1 else 'model plan'}", kern, plan))
        runs.append(("model", -1, 0))
        live = []
        for label, kern, plan in runs:
            laser_amd.set_option("f32_asm", 2 if kern >= 0 else 1)
            laser_amd.set_option("asm_kernel", kern)
            laser_amd.set_option("asm_plan", plan)
            C.fill_(float("nan"))
            rc = call()
            torch.cuda.synchronize()
            got = laser_amd.last_f32_asm()
            if rc != 0 or (kern >= 0 and got != kern + 1):
                rec["kernels"][label] = {"skipped": "not eligible"}
                continue
            info = {"kernel_index": got - 1, "wgs": laser_amd.get_option("last_asm_wgs"), "slices": laser_amd.get_option("last_asm_slices")}
            if mode == 0:
                if ref is None:
                    ref = C.clone()
                info["same_bits_as_first"] = bool(torch.equal(C, ref))
            else:
                if ref is None:
                    ref = C.clone()
                info["max_abs_diff_vs_first"] = float((C - ref).abs().max())
            rec["kernels"][label] = info
            live.append((label, kern, plan))
        times = {l: [] for l, _, _ in live}
        for label, kern, plan in live[:1]:
            laser_amd.set_option("f32_asm", 2 if kern >= 0 else 1)
            laser_amd.set_option("asm_kernel", kern)
            laser_amd.set_option("asm_plan", plan)
            warm(call)
        for _ in range(reps):
            for label, kern, plan in live:
                laser_amd.set_option("f32_asm", 2 if kern >= 0 else 1)
                laser_amd.set_option("asm_kernel", kern)
                laser_amd.set_option("asm_plan", plan)
                call()
                times[label].append(timed(call, fl))
        for label, ts in times.items():
            ts = sorted(ts)
            ms = ts[len(ts) // 2]
            rec["kernels"][label].update({"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / 157.3, 4), "min_ms": round(ts[0], 4)})
        best = min((l for l in times if l != "model"), key=lambda l: rec["kernels"][l]["ms"])
        rec["best_forced"] = best
        rec["model_vs_best_pct"] = round(100.0 * (rec["kernels"]["model"]["ms"] / rec["kernels"][best]["ms"] - 1.0), 2) if "ms" in rec["kernels"].get("model", {}) else None
        print(json.dumps(rec), flush=True)
laser_amd.set_option("asm_kernel", -1)
laser_amd.set_option("asm_plan", 0)
laser_amd.set_option("f32_asm", 1)
laser_amd.set_float_mode(0)
