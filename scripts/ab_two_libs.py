#!/usr/bin/env python3
"""Within-process A/B of two builds of the library: liblaser_hip.so vs an experimental build placed at
laser_amd/lib/liblaser_hip_exp.so (same sources with an extra -D...), interleaved rounds in ONE process
(cross-process variance on this pool is ~1-3 %).  Used in round 1 for: static s_setprio(1) on the
second-dispatched half of the 8-wave workgroups -> null (139.4 vs 139.3 TFLOP/s fast, 127.7 vs 127.6 laser-order)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {n: C.CDLL(os.path.join(root, "laser_amd", "lib", f)) for n, f in (("base", "liblaser_hip.so"), ("exp", "liblaser_hip_exp.so"))}
i64, vp = C.c_int64, C.c_void_p
for L in libs.values():
    L.laser_hip_gemm_strided_f32_dev.argtypes = [i64, i64, i64, C.c_float, vp, i64, i64, vp, i64, i64, C.c_float, vp, i64, i64, vp]
    L.laser_hip_set_float_mode.argtypes = [C.c_int]
n = 8192
A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; Cc = torch.zeros((n, n), device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(L): assert L.laser_hip_gemm_strided_f32_dev(n, n, n, 1.0, A.data_ptr(), n, 1, B.data_ptr(), n, 1, 0.0, Cc.data_ptr(), n, 1, st) == 0
for mode in (1, 0):
    res = {k: [] for k in libs}
    for L in libs.values(): L.laser_hip_set_float_mode(mode)
    for r in range(7):
        for name, L in libs.items():
            run(L); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): run(L)
            e1.record(); torch.cuda.synchronize()
            if r: res[name].append(e0.elapsed_time(e1) / 4)
    for name, v in res.items():
        v.sort(); print("fast " if mode else "laser", name, f"median {v[len(v)//2]:.4f} ms  min {v[0]:.4f}  -> {2*n**3/v[len(v)//2]/1e9:.1f} TF")
