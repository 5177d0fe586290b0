"""Raster group height (option asm_group_m: tile rows per group = which tiles share an XCD's L2) of the 64x64 kernels, plain plan:
per-launch time at a few mid sizes.  One JSON line per shape."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import laser_amd
from laser_amd import _lib as _lh
L = _lh.lib()
f = ctypes.c_float
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
laser_amd.set_option("f32_asm", 2); laser_amd.set_option("asm_plan", 1); laser_amd.set_option("asm_kernel", 12)
for n in (3072, 2304, 1920, 1536):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
    B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
    C = torch.zeros((n, n), device="cuda")
    call = lambda: L.laser_hip_gemm_strided_f32_dev(n, n, n, f(1.0), ctypes.c_void_p(A.data_ptr()), n, 1, ctypes.c_void_p(B.data_ptr()), n, 1, f(0.0), ctypes.c_void_p(C.data_ptr()), n, 1, stream)
    row = {"n": n, "kernel": "exact_64x64x32, one tile per workgroup"}
    for rnd in range(2):
        for gm in (0, 2, 4, 6, 8, 12, 16, 24, 48):
            laser_amd.set_option("asm_group_m", gm)
            for _ in range(6): call()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8): call()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 8)
            row[f"group_m={gm}" + ("" if rnd == 0 else " (again)")] = round(2.0 * n ** 3 / sorted(ts)[2] / 1e9, 1)
    print(json.dumps(row), flush=True)
laser_amd.set_option("asm_group_m", 0); laser_amd.set_option("asm_plan", 0); laser_amd.set_option("asm_kernel", -1); laser_amd.set_option("f32_asm", 1)
