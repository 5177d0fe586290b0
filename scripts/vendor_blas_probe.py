#!/usr/bin/env python3
"""Context only (nothing in the product calls a vendor library): torch.matmul -> rocBLAS / hipBLASLt on the same box, the
same operands and timing harness as bench_configs.py's C2 lines, next to laser_hip's two accumulation modes.
Round 6: laser_hip is timed through its C-ABI entry bound once with ctypes (what a binding does; the Python mirror's per-call
marshalling is ~10 us, a third of a 960^3 float64 launch -- the mirror's figure is kept beside it), and the three contenders are
timed in interleaved rounds (the first timing after a switch of library ran up to 15 % slow)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from laser_amd import _lib as _lh
from scripts.bench_configs import ev_time
torch.backends.cuda.matmul.allow_tf32 = False
L = _lh.lib()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
# usage: vendor_blas_probe.py [small]   ("small": the reference's published shapes and their neighbours -- VERDICT r4 next #5: a
# yardstick for the small-tile kernels, where no vendor number existed)
SHAPES = ((8192, torch.float32), (4096, torch.float32), (8192, torch.float64))
if len(sys.argv) > 1 and sys.argv[1] == "small":
    SHAPES = ((1024, torch.float32), (1536, torch.float32), (1920, torch.float32), (2560, torch.float32), (3072, torch.float32),
              (960, torch.float64), (1536, torch.float64), (1920, torch.float64))
for n, dt in SHAPES:
    A = (torch.rand((n, n), device="cuda", dtype=dt) - 0.5) * 0.2
    B = (torch.rand((n, n), device="cuda", dtype=dt) - 0.5) * 0.2
    C = torch.zeros((n, n), device="cuda", dtype=dt)
    f64 = dt == torch.float64
    ct = ctypes.c_double if f64 else ctypes.c_float
    fn = L.laser_hip_gemm_strided_f64_dev if f64 else L.laser_hip_gemm_strided_f32_dev
    cargs = (n, n, n, ct(1.0), ctypes.c_void_p(A.data_ptr()), n, 1, ctypes.c_void_p(B.data_ptr()), n, 1, ct(0.0), ctypes.c_void_p(C.data_ptr()), n, 1, stream)
    it = 5 if n >= 4096 else 9
    inner = 4 if n >= 4096 else 40
    runs = {"vendor_blas": (None, lambda: torch.matmul(A, B, out=C)), "laser_hip_laser_order": (0, lambda: fn(*cargs)), "laser_hip_fast": (1, lambda: fn(*cargs))}
    ts = {k: [] for k in runs}
    for rnd in range(3):
        for k, (mode, call) in runs.items():
            if mode is not None:
                laser_amd.set_float_mode(mode)
            med, mn = ev_time(call, iters=it, inner=inner)
            ts[k].append(med)
    rec = {"shape": n, "dtype": str(dt).replace("torch.", ""), "timed": "median of 3 interleaved rounds; laser_hip through its C-ABI entry bound once"}
    for k in runs:
        med = sorted(ts[k])[1]
        rec[f"{k}_ms"] = round(med, 4); rec[f"{k}_tflops"] = round(2.0 * n ** 3 / med / 1e9, 1)
    if n < 4096:
        for mode, name in ((0, "laser_order"), (1, "fast")):
            laser_amd.set_float_mode(mode)
            med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=it, inner=inner)
            rec[f"laser_hip_{name}_python_mirror_ms"] = round(med, 4)
    laser_amd.set_float_mode(0)
    print(json.dumps(rec), flush=True)
