#!/usr/bin/env python3
"""Host-pointer gemm_strided end to end at 8192^3 (H2D A, B + kernels + D2H C): the 2-D (row x column panel) pipeline
vs the row-panel pipeline, pageable and pinned host memory.  One JSON line per arm; results compared bit for bit."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import laser_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(3)
Ah = rng.uniform(-0.1, 0.1, (n, n)).astype(np.float32)
Bh = rng.uniform(-0.1, 0.1, (n, n)).astype(np.float32)
Ch = np.zeros((n, n), np.float32)
Ap, Bp, Cp = laser_amd.pinned_host_buffer((n, n)), laser_amd.pinned_host_buffer((n, n)), laser_amd.pinned_host_buffer((n, n))
Ap[:] = Ah; Bp[:] = Bh
ref = None
for mem, (A, B, C) in (("pageable", (Ah, Bh, Ch)), ("pinned", (Ap, Bp, Cp))):
    for mode in (1, 0):
        laser_amd.set_host_pipeline(mode)
        laser_amd.matmul(A, B, 1, 0, C)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); laser_amd.matmul(A, B, 1, 0, C); ts.append(time.perf_counter() - t0)
        ts.sort()
        if ref is None:
            ref = C.copy()
        assert np.array_equal(ref, C)
        print(json.dumps({"config": f"fp32 {n}^3 host-pointer end-to-end", "memory": mem, "pipeline": "rows x columns" if mode else "rows only",
                          "ms_med": round(ts[1] * 1e3, 2), "ms_min": round(ts[0] * 1e3, 2), "tflops": round(2.0 * n ** 3 / ts[1] / 1e12, 2)}), flush=True)
laser_amd.set_host_pipeline(1)
