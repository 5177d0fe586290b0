#!/usr/bin/env python3
"""BASELINE configs[2] (fp32 4096^3, B passed transposed: rowStrideB = 1, colStrideB = K) through the device-resident entry point,
repeated, for rocprofv3 (scripts/gpu_profile_cmd.sh c3 python scripts/c3_run.py) and a quick timing.   usage: c3_run.py [iters] [full]
"full": the full form of the config -- A an every-2nd-row view, B transposed, C with column stride 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, laser_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
full = len(sys.argv) > 2 and sys.argv[2] == "full"
n = 4096
g = torch.Generator(device="cuda").manual_seed(5)
Abig = torch.rand((2 * n, n), generator=g, device="cuda") * 0.2 - 0.1
Bt = torch.rand((n, n), generator=g, device="cuda") * 0.2 - 0.1
Cbuf = torch.zeros((n, 2 * n), device="cuda")
A, B, C = (Abig[::2], Bt.t(), Cbuf[:, ::2]) if full else (Abig[:n], Bt.t(), Cbuf[:, :n].contiguous())
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    for _ in range(60):      # (the clocks ramp up over the first ~50 ms of work after idle: warm-only averages drop these)
        laser_amd.matmul(A, B, 1, 0, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        laser_amd.matmul(A, B, 1, 0, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{'laser' if mode == 0 else 'fast '} asm_kernel={laser_amd.last_f32_asm()} {ms:.4f} ms {2.0 * n ** 3 / ms / 1e9:6.1f} TF", flush=True)
laser_amd.set_float_mode(0)
