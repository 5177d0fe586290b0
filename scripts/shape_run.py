#!/usr/bin/env python3
"""One GEMM shape, repeated (for rocprofv3 passes and quick timings): shape_run.py M N K [mode 0|1] [asm_kernel] [asm_plan] [reps] [f32|f64]"""
import ctypes
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laser_amd
from laser_amd import _lib as _lh
M, N, K = (int(x) for x in sys.argv[1:4])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
kern = int(sys.argv[5]) if len(sys.argv) > 5 else -1
plan = int(sys.argv[6]) if len(sys.argv) > 6 else 0
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 50
f64 = len(sys.argv) > 8 and sys.argv[8] == "f64"
dt = torch.float64 if f64 else torch.float32
L = _lh.lib()
fn = L.laser_hip_gemm_strided_f64_dev if f64 else L.laser_hip_gemm_strided_f32_dev
ct = ctypes.c_double if f64 else ctypes.c_float
g = torch.Generator(device="cuda").manual_seed(1)
A = ((torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2).to(dt)
B = ((torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2).to(dt)
C = torch.zeros((M, N), device="cuda", dtype=dt)
laser_amd.set_float_mode(mode)
laser_amd.set_option("f64_asm" if f64 else "f32_asm", 2)
laser_amd.set_option("asm_kernel", kern)
laser_amd.set_option("asm_plan", plan)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
cargs = (M, N, K, ct(1.0), ctypes.c_void_p(A.data_ptr()), K, 1, ctypes.c_void_p(B.data_ptr()), N, 1, ct(0.0), ctypes.c_void_p(C.data_ptr()), N, 1, stream)
for _ in range(10):
    fn(*cargs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn(*cargs)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps({"M": M, "N": N, "K": K, "mode": mode, "asm_kernel": laser_amd.get_option("last_f64_asm") if f64 else laser_amd.last_f32_asm(), "wgs": laser_amd.get_option("last_asm_wgs"),
                  "slices": laser_amd.get_option("last_asm_slices"), "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}))
