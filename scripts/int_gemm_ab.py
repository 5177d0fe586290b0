#!/usr/bin/env python3
"""int32 / int64 GEMM (int8-limb decomposition on the matrix cores): the hand-scheduled kernels (option i32_asm, default) vs the
compiler-scheduled limb kernel, packing pass included in both; bit-identity per line.  One JSON line per shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from scripts.bench_configs import ev_time
for dt, n in [(torch.int32, 1920), (torch.int32, 2048), (torch.int32, 4096), (torch.int32, 8192), (torch.int64, 960), (torch.int64, 2048), (torch.int64, 4096), (torch.int64, 8192)]:
    lim = 2**31 if dt == torch.int32 else 2**62
    A = torch.randint(-lim, lim - 1, (n, n), device="cuda", dtype=dt)
    B = torch.randint(-lim, lim - 1, (n, n), device="cuda", dtype=dt)
    rec = {"dtype": str(dt).split(".")[1], "shape": [n, n, n]}
    outs = {}
    for asm in (1, 0):
        laser_amd.set_option("i32_asm", asm)
        C = torch.zeros((n, n), device="cuda", dtype=dt)
        ms, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7)
        key = "asm" if asm else "compiler"
        rec[key + "_ms"] = round(ms, 4)
        rec[key + "_tintops"] = round(2.0 * n ** 3 / ms / 1e9, 1)
        rec[key + "_used_asm"] = laser_amd.get_option("last_i32_asm")
        outs[asm] = C
    rec["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
    print(json.dumps(rec), flush=True)
laser_amd.set_option("i32_asm", 1)
