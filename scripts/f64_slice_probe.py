#!/usr/bin/env python3
"""float64 slice-parallel eligibility probe: laser-order dgemm on few-tile shapes with the slice-parallel path forced
(tile threshold 100000, minimum 2 slices) vs off.  Bit-identity of the two is asserted.  One JSON line per shape."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from scripts.bench_configs import ev_time

shapes = [(960, 960, 960), (512, 512, 2048), (768, 768, 768), (1024, 1024, 1024), (1280, 1280, 1280), (1920, 1920, 1920),
          (640, 640, 4096), (2048, 2048, 2048)]
laser_amd.set_float_mode(0)
for (M, N, K) in shapes:
    A = (torch.rand((M, K), device="cuda", dtype=torch.float64) - 0.5) * 0.2
    B = (torch.rand((K, N), device="cuda", dtype=torch.float64) - 0.5) * 0.2
    C0 = torch.zeros((M, N), device="cuda", dtype=torch.float64)
    C1 = torch.zeros((M, N), device="cuda", dtype=torch.float64)
    laser_amd.set_option("slice_parallel", 0)
    off, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C0), iters=9)
    laser_amd.set_option("slice_parallel", 1); laser_amd.set_option("slice_parallel_tiles", 100000); laser_amd.set_option("slice_parallel_min", 2)
    on, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C1), iters=9)
    laser_amd.set_option("slice_parallel_tiles", 0)
    dflt, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C1), iters=9)
    tiles64 = ((M + 63) // 64) * ((N + 63) // 64)
    print(json.dumps({"shape": [M, N, K], "tiles64": tiles64, "slices": (K + 255) // 256, "off_ms": round(off, 4), "forced_ms": round(on, 4),
                      "speedup": round(off / on, 3), "bit_identical": bool(torch.equal(C0, C1)),
                      "tflops_off": round(2.0 * M * N * K / off / 1e9, 2), "tflops_forced": round(2.0 * M * N * K / on / 1e9, 2),
                      "default_ms": round(dflt, 4), "tflops_default": round(2.0 * M * N * K / dflt / 1e9, 2)}), flush=True)
