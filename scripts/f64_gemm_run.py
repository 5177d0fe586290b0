#!/usr/bin/env python3
"""float64 GEMM: the hand-scheduled kernels (option f64_asm, default) vs the compiler-scheduled f64 MFMA kernels, both
accumulation modes, the reference's f64 bench shape 960^3 (benchmarks/gemm/gemm_bench_float64.nim) first; bit-identity of
the two is checked per line.  One JSON line per shape and mode."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from scripts.bench_configs import ev_time
PEAK = 78.6
shapes = [(960,) * 3, (1024,) * 3, (1536,) * 3, (1920,) * 3, (2048,) * 3, (3072,) * 3, (4096,) * 3, (8192,) * 3] if len(sys.argv) < 2 else [tuple(int(x) for x in sys.argv[1:4])]
for (M, N, K) in shapes:
    A = (torch.rand((M, K), device="cuda", dtype=torch.float64) - 0.5) * 0.2
    B = (torch.rand((K, N), device="cuda", dtype=torch.float64) - 0.5) * 0.2
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"shape": [M, N, K], "mode": "laser_order" if mode == 0 else "fast"}
        outs = {}
        for asm in (1, 0):
            laser_amd.set_option("f64_asm", asm)
            C = torch.zeros((M, N), device="cuda", dtype=torch.float64)
            ms, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7)
            key = "asm" if asm else "compiler"
            rec[key + "_ms"] = round(ms, 4)
            rec[key + "_tflops"] = round(2.0 * M * N * K / ms / 1e9, 2)
            if asm:
                rec["asm_kernel"] = laser_amd.get_option("last_f64_asm")
            outs[asm] = C
        rec["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
        rec["frac_f64_peak"] = round(rec["asm_tflops"] / PEAK, 4)
        print(json.dumps(rec), flush=True)
laser_amd.set_option("f64_asm", 1); laser_amd.set_float_mode(0)
