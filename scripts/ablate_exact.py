#!/usr/bin/env python3
"""Price the parts of the f32 main loop with COMPILE-TIME ablations of the production kernels: experimental builds
laser_amd/lib/liblaser_hip_dbg<mask>.so (make ... CXXFLAGS+=-DLH_DBG_MASK=<mask>: 1 = no HBM loads of later tiles, 2 = no
LDS stores, 4 = no mid-tile barrier) against liblaser_hip.so, interleaved rounds in one process, 8192^3.  Results of the
ablated builds are wrong by construction; this only times."""
import ctypes as C, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
libs = {"full": C.CDLL(os.path.join(ROOT, "laser_amd", "lib", "liblaser_hip.so"))}
for f in sorted(glob.glob(os.path.join(ROOT, "laser_amd", "lib", "liblaser_hip_dbg*.so"))):
    libs["mask" + re.search(r"dbg(\d+)", f).group(1)] = C.CDLL(f)
what = {"full": "production", "mask1": "no HBM loads", "mask2": "no LDS stores", "mask3": "no loads + stores", "mask4": "no barrier",
        "mask7": "MFMA + fragment reads only"}
i64, vp = C.c_int64, C.c_void_p
for L in libs.values():
    L.laser_hip_gemm_strided_f32_dev.argtypes = [i64, i64, i64, C.c_float, vp, i64, i64, vp, i64, i64, C.c_float, vp, i64, i64, vp]
    L.laser_hip_set_float_mode.argtypes = [C.c_int]; L.laser_hip_set_f32_config.argtypes = [C.c_int]
n = 8192
A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; Cc = torch.zeros((n, n), device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(L): assert L.laser_hip_gemm_strided_f32_dev(n, n, n, 1.0, A.data_ptr(), n, 1, B.data_ptr(), n, 1, 0.0, Cc.data_ptr(), n, 1, st) == 0
CASES = ((0, 1, "256x256x16 8 waves, fast"), (4, 0, "256x128x32 8 waves, laser-order"), (4, 1, "256x128x32 8 waves, fast"))
if len(sys.argv) > 1 and sys.argv[1] == "dma":      # the LDS-DMA kernel's own variants (cfg -1: the library's dispatch)
    CASES = ((-1, 0, "LDS-DMA 256x128x32, laser-order"), (-1, 1, "LDS-DMA 256x128x32, fast"))
    what.update(mask1="DMA pieces late in each group", mask2="no DMA requests (MFMA + fragment reads + selects only)")
for cfg, mode, label in CASES:
    res = {k: [] for k in libs}
    for L in libs.values():
        L.laser_hip_set_float_mode(mode); L.laser_hip_set_f32_config(cfg)
    for r in range(6):
        for name, L in libs.items():
            run(L); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): run(L)
            e1.record(); torch.cuda.synchronize()
            if r: res[name].append(e0.elapsed_time(e1) / 4)
    for name, v in res.items():
        v.sort(); med = v[len(v) // 2]
        print(json.dumps({"kernel": label, "variant": what.get(name, name), "ms": round(med, 4), "tflops": round(2.0 * n ** 3 / med / 1e9, 1),
                          "frac_mfma_peak": round(2.0 * n ** 3 / med / 1e9 / 157.3, 4)}), flush=True)
