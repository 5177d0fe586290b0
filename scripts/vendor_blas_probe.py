#!/usr/bin/env python3
"""Context only (nothing in the product calls a vendor library): torch.matmul -> rocBLAS / hipBLASLt on the same box, the
same operands and timing harness as bench_configs.py's C2 lines, next to laser_hip's two accumulation modes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from scripts.bench_configs import ev_time
torch.backends.cuda.matmul.allow_tf32 = False
# usage: vendor_blas_probe.py [small]   ("small": the reference's published shapes and their neighbours -- VERDICT r4 next #5: a
# yardstick for the small-tile kernels, where no vendor number existed)
SHAPES = ((8192, torch.float32), (4096, torch.float32), (8192, torch.float64))
if len(sys.argv) > 1 and sys.argv[1] == "small":
    SHAPES = ((1024, torch.float32), (1536, torch.float32), (1920, torch.float32), (2560, torch.float32), (3072, torch.float32),
              (960, torch.float64), (1536, torch.float64))
for n, dt in SHAPES:
    A = (torch.rand((n, n), device="cuda", dtype=dt) - 0.5) * 0.2
    B = (torch.rand((n, n), device="cuda", dtype=dt) - 0.5) * 0.2
    C = torch.zeros((n, n), device="cuda", dtype=dt)
    med, mn = ev_time(lambda: torch.matmul(A, B, out=C), iters=7 if n >= 4096 else 40)
    rec = {"shape": n, "dtype": str(dt).replace("torch.", ""), "vendor_blas_ms": round(med, 4), "vendor_blas_tflops": round(2.0 * n ** 3 / med / 1e9, 1)}
    for mode, name in ((0, "laser_order"), (1, "fast")):
        laser_amd.set_float_mode(mode)
        med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7 if n >= 4096 else 40)
        rec[f"laser_hip_{name}_ms"] = round(med, 4); rec[f"laser_hip_{name}_tflops"] = round(2.0 * n ** 3 / med / 1e9, 1)
    laser_amd.set_float_mode(0)
    print(json.dumps(rec), flush=True)
