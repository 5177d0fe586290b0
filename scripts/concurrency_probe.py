#!/usr/bin/env python3
"""Concurrent cut launches: which combination of host threads / streams / plans ever makes a receiver give up waiting
(option asm_fixup_timeouts) or produces a wrong result.  One JSON line per variation."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import laser_amd

rng = np.random.default_rng(2024)
shapes = [(1100, 900, 1600), (700, 1300, 2100), (1536, 1536, 1100), (520, 2050, 1030)]
probs = []
laser_amd.set_float_mode(0)
laser_amd.set_option("slice_parallel", 0)
laser_amd.set_option("f32_asm", 2)
for (M, N, K) in shapes:
    A = torch.from_numpy(rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32)).cuda()
    laser_amd.set_option("asm_plan", 1)
    want = laser_amd.matmul(A, B).clone()
    probs.append((A, B, want))
torch.cuda.synchronize()


def variation(name, nthreads, nstreams, plans, reps=8, kernel=-1):
    errors = []
    t0 = time.perf_counter()
    laser_amd.set_option("asm_kernel", kernel)

    def worker(tid):
        torch.cuda.set_device(0)
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        outs = []
        for rep in range(reps):
            for si, st in enumerate(streams):
                A, B, want = probs[(2 * tid + si + rep) % len(probs)]
                with torch.cuda.stream(st):
                    if len(plans) > 1:
                        laser_amd.set_option("asm_plan", plans[(rep + si) % len(plans)])
                    outs.append((laser_amd.matmul(A, B), want, (tid, rep, si)))
        for st in streams:
            st.synchronize()
        for C, want, tag in outs:
            if not torch.equal(C, want):
                errors.append(tag)

    laser_amd.set_option("asm_plan", plans[0])
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    rec = {"variation": name, "threads": nthreads, "streams": nstreams, "plans": plans, "kernel": kernel, "wrong": len(errors), "first_wrong": errors[:3],
           "timeouts_total": laser_amd.get_option("asm_fixup_timeouts"), "seconds": round(time.perf_counter() - t0, 2)}
    print(json.dumps(rec), flush=True)


for rnd in range(3):
    variation("1 thread, 1 stream, persistent", 1, 1, [2])
    variation("1 thread, 2 streams, auto", 1, 2, [0])
    variation("1 thread, 2 streams, persistent", 1, 2, [2])
    variation("2 threads, 1 stream each, persistent", 2, 1, [2])
    variation("2 threads, 2 streams each, persistent", 2, 2, [2])
    variation("2 threads, 2 streams each, auto", 2, 2, [0])
    variation("2 threads, 2 streams each, mixed", 2, 2, [2, 0])
    variation("1 thread, 2 streams, persistent, 64x64 only", 1, 2, [2], kernel=12)
    variation("1 thread, 2 streams, persistent, 256x128 only", 1, 2, [2], kernel=0)
    variation("1 thread, 4 streams, persistent, 128x128 only", 1, 4, [2], kernel=2)
laser_amd.set_option("asm_plan", 0); laser_amd.set_option("asm_kernel", -1); laser_amd.set_option("f32_asm", 1); laser_amd.set_option("slice_parallel", 1)
