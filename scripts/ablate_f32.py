#!/usr/bin/env python3
"""Price the parts of the f32 MFMA main loop on the GPU box: the production kernel template built with run-time
ablation switches (build/liblaser_probe.so, `make -C scripts/probes`) -- skip HBM loads / LDS stores / barriers.
Results are wrong by construction when a switch is on; this only times."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = C.CDLL(os.path.join(ROOT, "build", "liblaser_probe.so"))
L.laser_probe_f32.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
Cc = torch.zeros((n, n), device="cuda")
names = {0: "256x256x16", 1: "256x128x32", 2: "256x128x16", 3: "128x128x16"}
what = {0: "full", 1: "no HBM loads", 2: "no LDS stores", 3: "no loads+stores", 4: "no barrier", 7: "MFMA + LDS reads only"}
st = torch.cuda.current_stream().cuda_stream
for shape, exact in [(0, 0), (1, 0), (1, 1), (2, 1), (3, 1), (3, 0)]:
    for dbg in (0, 1, 2, 3, 4, 7):
        def run():
            rc = L.laser_probe_f32(n, A.data_ptr(), B.data_ptr(), Cc.data_ptr(), dbg, shape, exact, st)
            assert rc == 0, rc
        for _ in range(3): run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
        ts.sort()
        print(f"{names[shape]:12s} {'laser' if exact else 'fast ':5s} dbg={dbg} {what[dbg]:24s} {ts[2]:.4f} ms {2*n**3/ts[2]/1e9:7.1f} TF", flush=True)
