#!/usr/bin/env python3
"""Price the parts of the f32 main loop with the probe kernel's ablation switches (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
L = laser_amd.lib()
L.laser_hip_probe_f32_dev.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
n = 8192
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
Cc = torch.zeros((n, n), device="cuda")
base = {0: "full(probe build)", 3: "no loads+stores", 4: "no barrier", 7: "MFMA + LDS reads only"}
shapes = {0: "256x256x16 w128x64", 1: "256x128x32 w64x64", 2: "256x128x16 w64x64"}
names = {(sh << 8) | d: f"{shapes[sh]:20s} {n}" for sh in shapes for d, n in base.items()}
res = {k: [] for k in names}
for r in range(4):
    for dbg in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            rc = L.laser_hip_probe_f32_dev(n, A.data_ptr(), B.data_ptr(), Cc.data_ptr(), dbg, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, L.laser_hip_last_error()
        e1.record(); torch.cuda.synchronize()
        if r: res[dbg].append(e0.elapsed_time(e1) / 3)
for dbg, v in res.items():
    v.sort(); ms = v[len(v)//2]
    print(f"{names[dbg]:46s} {ms:.4f} ms  {2*n**3/ms/1e9:.1f} TFLOP/s")
