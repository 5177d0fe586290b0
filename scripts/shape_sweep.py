#!/usr/bin/env python3
"""Mid-size / ragged shapes: library default (assembly kernels where they apply) vs assembly kernels off,
both accumulation modes.  One JSON line per shape and mode -> profiles/r03/shape_sweep_*.jsonl."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
shapes = [(512,) * 3, (768,) * 3, (1024,) * 3, (1280,) * 3, (1536,) * 3, (1920,) * 3, (2048,) * 3, (3072,) * 3, (4096,) * 3, (6144,) * 3, (4100,) * 3, (1000, 3000, 2000), (4095, 4097, 4099),
          (2048, 8192, 1024), (8192, 8192, 512)]
for (M, N, K) in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2
    B = (torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2
    if len(sys.argv) > 1 and sys.argv[1] == "nt":          # B passed transposed (BASELINE configs[2])
        B = B.t().contiguous().t()
    C = torch.zeros((M, N), device="cuda")
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        rec = {"M": M, "N": N, "K": K, "mode": "laser_order" if mode == 0 else "fast"}
        ref = None
        for asm in (1, 0):
            laser_amd.set_f32_asm(asm)
            for _ in range(max(3, min(400, int(0.03 / max(1e-6, 2.0 * M * N * K / 100e12))))):   # clocks up: ~30 ms of work
                laser_amd.matmul(A, B, 1, 0, C)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    laser_amd.matmul(A, B, 1, 0, C)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            ms = sorted(ts)[2]
            key = "asm" if asm else "compiler"
            rec[key + "_ms"] = round(ms, 4)
            rec[key + "_tflops"] = round(2.0 * M * N * K / ms / 1e9, 1)
            if asm:
                rec["asm_kernel"] = laser_amd.last_f32_asm()
                ref = C.clone()
            else:
                rec["bit_identical"] = bool(torch.equal(ref, C))
        rec["frac_mfma_peak"] = round(rec["asm_tflops"] / 157.3, 4)
        print(json.dumps(rec), flush=True)
laser_amd.set_float_mode(0); laser_amd.set_f32_asm(1)
