#!/usr/bin/env python3
"""Explicit im2col alone (the public im2col*[T] entry point, HBM-bound): device-resident, C-ABI symbol bound once, 16 launches per
sample.  Bytes = input read once + workspace written once (an explicit im2col is priced against HBM, not MFMA)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
from scripts.bench_configs import ev_time
L = laser_amd.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, ishape, k, pad, stv, dt in (("C4 (32,128,56,56) 3x3 pad 1 f32", (32, 128, 56, 56), (3, 3), (1, 1), (1, 1), torch.float32),
                                      ("C4' (32,128,56,56) 3x3 pad 0 f32", (32, 128, 56, 56), (3, 3), (0, 0), (1, 1), torch.float32),
                                      ("reference conv bench (16,3,224,224) 3x3 pad 0 f32", (16, 3, 224, 224), (3, 3), (0, 0), (1, 1), torch.float32),
                                      ("(16,64,112,112) 3x3 pad 1 stride 2 f32", (16, 64, 112, 112), (3, 3), (1, 1), (2, 2), torch.float32),
                                      ("(8,128,56,56) 3x3 pad 1 f64", (8, 128, 56, 56), (3, 3), (1, 1), (1, 1), torch.float64),
                                      ("(8,32,27,27) 5x5 pad 2 f32 (oW % 4 != 0)", (8, 32, 27, 27), (5, 5), (2, 2), (1, 1), torch.float32)):
    n, c, h, w = ishape
    oh = (h + 2 * pad[0] - k[0]) // stv[0] + 1
    ow = (w + 2 * pad[1] - k[1]) // stv[1] + 1
    x = torch.rand(ishape, device="cuda", dtype=dt)
    ws = torch.empty((n, c * k[0] * k[1], oh * ow), device="cuda", dtype=dt)
    fn = L.laser_hip_im2col_f32_dev if dt == torch.float32 else L.laser_hip_im2col_f64_dev
    args = (ctypes.c_void_p(ws.data_ptr()), oh, ow, ctypes.c_void_p(x.data_ptr()), n, c, h, w, k[0], k[1], pad[0], pad[1], stv[0], stv[1], st)
    assert fn(*args) == 0
    med, mn = ev_time(lambda: fn(*args), iters=9, inner=16)
    byts = (x.numel() + ws.numel()) * x.element_size()
    sweep = {}
    if len(sys.argv) > 1 and sys.argv[1] == "bands":
        for band in (256, 512, 784, 1568, 3136, 6272):
            laser_amd.set_option("im2col_band", band)
            m_, _ = ev_time(lambda: fn(*args), iters=7, inner=16)
            sweep[str(band)] = round((x.numel() + ws.numel()) * x.element_size() / (m_ * 1e-3) / 1e9, 1)
        laser_amd.set_option("im2col_band", 0)
    print(json.dumps({"config": "im2col alone " + name, "gbps_by_band_pixels": sweep, "ms_med": round(med, 4), "ms_min": round(mn, 4), "bytes": byts,
                      "gbps": round(byts / (med * 1e-3) / 1e9, 1), "frac_hbm_peak": round(byts / (med * 1e-3) / 1e9 / 8000.0, 4)}), flush=True)
