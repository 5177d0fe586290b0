#!/usr/bin/env python3
"""The explicit im2col of C4 (32,128,56,56) 3x3 pad 1, repeated, for rocprofv3 (gpu_profile_cmd.sh im2col python scripts/im2col_c4_run.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
L = laser_amd.lib()
x = torch.rand((32, 128, 56, 56), device="cuda")
ws = torch.empty((32, 128 * 9, 56 * 56), device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
args = (ctypes.c_void_p(ws.data_ptr()), 56, 56, ctypes.c_void_p(x.data_ptr()), 32, 128, 56, 56, 3, 3, 1, 1, 1, 1, st)
for _ in range(100):
    assert L.laser_hip_im2col_f32_dev(*args) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    L.laser_hip_im2col_f32_dev(*args)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 100
print(f"im2col C4: {ms:.4f} ms, {(x.numel() + ws.numel()) * 4 / ms / 1e9:.2f} TB/s (algorithmic bytes {(x.numel() + ws.numel()) * 4})")
