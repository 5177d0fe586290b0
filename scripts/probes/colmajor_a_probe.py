"""A column-major (rowStrideA = 1, colStrideA = M) x B row-major at 4096^3 through the C-ABI: per-call time in both accumulation modes
next to the contiguous product -- and, under `rocprofv3 --kernel-trace --stats`, the duration of every kernel the call launches."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import laser_amd
from laser_amd import _lib as _lh
L = _lh.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
At = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2      # K x M row-major = A column-major
A = At.t().contiguous()
B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
C = torch.zeros((n, n), device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f = ctypes.c_float
def call(col):
    if col: L.laser_hip_gemm_strided_f32_dev(n, n, n, f(1.0), ctypes.c_void_p(At.data_ptr()), 1, n, ctypes.c_void_p(B.data_ptr()), n, 1, f(0.0), ctypes.c_void_p(C.data_ptr()), n, 1, stream)
    else: L.laser_hip_gemm_strided_f32_dev(n, n, n, f(1.0), ctypes.c_void_p(A.data_ptr()), n, 1, ctypes.c_void_p(B.data_ptr()), n, 1, f(0.0), ctypes.c_void_p(C.data_ptr()), n, 1, stream)
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    row = {"n": n, "mode": "laser_order" if mode == 0 else "fast"}
    for col in (0, 1):
        for _ in range(10): call(col)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call(col)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
        row["column_major_a_ms" if col else "contiguous_ms"] = round(sorted(ts)[2], 4)
        ts = []
        for _ in range(7):        # the pattern of scripts/bench_configs.py: 4 calls, then a synchronise
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): call(col)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
        row["column_major_a_4_per_sync_ms" if col else "contiguous_4_per_sync_ms"] = round(sorted(ts)[3], 4)
        row["kernel_col" if col else "kernel"] = laser_amd.last_f32_asm()
    print(json.dumps(row), flush=True)
laser_amd.set_float_mode(0)
