#!/usr/bin/env python3
"""PCIe rates the host-pointer pipeline lives on: blocking H2D / D2H of 256 MB from pinned and pageable host memory
(laser_hip_storage_upload / _download = hipMemcpy), alone and both directions at once (two host threads)."""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import laser_amd
from laser_amd import _lib
L = _lib.lib()
n = 8192
dev = [torch.empty((n, n), dtype=torch.float32, device="cuda") for _ in range(2)]
bufs = {"pinned": [laser_amd.pinned_host_buffer((n, n)) for _ in range(2)], "pageable": [np.zeros((n, n), np.float32) for _ in range(2)]}
for v in bufs.values():
    for b in v: b[:] = 1.0
def up(h, d): _lib.check(L.laser_hip_storage_upload(C.c_void_p(d.data_ptr()), C.c_void_p(h.ctypes.data), h.nbytes))
def down(h, d): _lib.check(L.laser_hip_storage_download(C.c_void_p(h.ctypes.data), C.c_void_p(d.data_ptr()), h.nbytes))
def timeit(fn, reps=4):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
gb = n * n * 4 / 1e9
for kind, (h0, h1) in bufs.items():
    tu = timeit(lambda: up(h0, dev[0])); td = timeit(lambda: down(h1, dev[1]))
    def both():
        t = threading.Thread(target=lambda: down(h1, dev[1])); t.start(); up(h0, dev[0]); t.join()
    tb = timeit(both)
    # the same with asynchronous copies on two separate streams (what the pipeline's upload / download sides use)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    def up_s(): _lib.check(L.laser_hip_storage_upload_stream(C.c_void_p(dev[0].data_ptr()), C.c_void_p(h0.ctypes.data), h0.nbytes, C.c_void_p(s0.cuda_stream)))
    def down_s(): _lib.check(L.laser_hip_storage_download_stream(C.c_void_p(h1.ctypes.data), C.c_void_p(dev[1].data_ptr()), h1.nbytes, C.c_void_p(s1.cuda_stream)))
    def both_s():
        t = threading.Thread(target=down_s); t.start(); up_s(); t.join()
    tbs = timeit(both_s)
    print(json.dumps({"memory": kind, "h2d_gbps": round(gb / tu, 1), "d2h_gbps": round(gb / td, 1), "both_ms": round(tb * 1e3, 2),
                      "both_gbps_each": round(gb / tb, 1), "both_two_streams_ms": round(tbs * 1e3, 2),
                      "both_two_streams_gbps_each": round(gb / tbs, 1)}), flush=True)
