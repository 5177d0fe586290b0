#!/usr/bin/env python3
"""Context only (nothing in the product calls a vendor library): square fp32 products n = 1024 .. 6144 in steps of 256 -- the library's
own choice of tile and launch plan in both accumulation modes, beside torch.matmul (the vendor BLAS) on the same operands, interleaved.
One JSON line per size.  usage: size_sweep_vendor.py [first] [last] [step]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
import bench
torch.backends.cuda.matmul.allow_tf32 = False
a0, a1, st = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 1024), (2, 6144), (3, 256)))


def timed(call, flops):
    inner = max(4, min(64, int(3e-3 / max(1e-6, flops / 100e12))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


for n in range(a0, a1 + 1, st):
    A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2
    B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2
    C = torch.zeros((n, n), device="cuda")
    fl = 2.0 * n ** 3
    calls = {"vendor": lambda: torch.matmul(A, B, out=C)}
    for mode, name in ((0, "laser_order"), (1, "fast")):
        def f(mode=mode):
            laser_amd.set_float_mode(mode)
            laser_amd.matmul(A, B, 1, 0, C)
        calls[name] = f
    rec = {"n": n}
    for name, f in calls.items():
        for _ in range(max(3, min(200, int(0.02 / max(1e-6, fl / 100e12))))):
            f()
        torch.cuda.synchronize()
        if name != "vendor":
            k = laser_amd.last_f32_asm()
            rec[name + "_kernel"] = bench.ASM_KERNEL_SYMBOLS[k - 1] if k else "compiler-scheduled"
            rec[name + "_plan"] = [laser_amd.get_option("last_asm_wgs"), laser_amd.get_option("last_asm_slices"), laser_amd.get_option("last_asm_rem")]      # (workgroups, K slices, tiles left to the K-cut launch of a hybrid plan)
    ts = {k: [] for k in calls}
    for _ in range(5):
        for name, f in calls.items():
            f()
            ts[name].append(timed(f, fl))
    for name in calls:
        v = sorted(ts[name])
        rec[name + "_tflops"] = round(fl / v[len(v) // 2] / 1e9, 1)
    rec["laser_order_vs_vendor_pct"] = round(100.0 * (rec["laser_order_tflops"] / rec["vendor_tflops"] - 1.0), 1)
    rec["fast_vs_vendor_pct"] = round(100.0 * (rec["fast_tflops"] / rec["vendor_tflops"] - 1.0), 1)
    print(json.dumps(rec), flush=True)
laser_amd.set_float_mode(0)
