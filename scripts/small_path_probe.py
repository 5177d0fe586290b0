#!/usr/bin/env python3
"""Where does the small-matrix kernel (one wave per block of C) beat the tiled / slice-parallel kernels?  Device-resident
launch time per shape with the small path on and off (the dispatch rule of gemm_small.hip admits <= 256 blocks, K <= 1024).
Timed with events over back-to-back launches issued from a C-level loop-free Python caller, so very small times are
host-bound: tests/cpp/small_gemm_bench.cpp is the authoritative number for 128^3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
shapes = [(32, 32, 32), (64, 64, 64), (128, 128, 128), (128, 128, 512), (128, 128, 1024), (256, 256, 128), (256, 256, 256), (256, 256, 1024),
          (384, 384, 384), (512, 512, 128), (512, 512, 512), (512, 512, 1024), (500, 500, 300), (64, 2048, 256), (2048, 64, 256), (16, 16, 1024)]
for (M, N, K) in shapes:
    A = (torch.rand((M, K), device="cuda") - 0.5) * 0.2
    B = (torch.rand((K, N), device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    res = {}
    for on in (1, 0):
        laser_amd.set_small_path(on)
        fn = lambda: laser_amd.matmul(A, B, 1, 0, C)
        for _ in range(20): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        ts.sort(); res[on] = (ts[2], laser_amd.last_f32_config())
    print(f"{M}x{N}x{K}: small {res[1][0]:8.2f} us (path {res[1][1]})   tiled {res[0][0]:8.2f} us (cfg {res[0][1]})   ratio {res[1][0]/res[0][0]:.2f}", flush=True)
laser_amd.set_small_path(1)
