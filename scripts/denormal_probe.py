#!/usr/bin/env python3
"""Do the laser-order kernels keep Laser's bits when products, chain sums and folds fall into the float32 subnormal range?  (The oracle is
an fmaf chain on the host: subnormals are kept.)  Operands scaled so that products are ~1e-40 .. 1e-38; every tile family forced; also
the float64 kernels at ~1e-310.  One JSON line per case.  usage: denormal_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import laser_amd
from oracle import oracle
oracle.build()
rng = np.random.default_rng(5)
M, N, K = 640, 768, 1100
for dt, scales in ((np.float32, (1.0, 1e-19, 3e-20, 1e-20)), (np.float64, (1.0, 1e-154, 3e-155))):
    for sc in scales:
        A = (rng.uniform(-1, 1, (M, K)) * sc).astype(dt)
        B = (rng.uniform(-1, 1, (K, N)) * sc).astype(dt)
        want = oracle.matmul(A, B)
        tiny = np.finfo(dt).tiny
        frac_sub = float(np.mean((np.abs(want) < tiny) & (want != 0)))
        dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        kerns = (-1, 0, 2, 12, 30, 46, 50, 54, 58, 62) if dt == np.float32 else (-1,)
        for kern in kerns:
            laser_amd.set_option("f32_asm", 2 if kern >= 0 else 1)
            laser_amd.set_option("asm_kernel", kern)
            laser_amd.set_option("asm_plan", 1 if kern >= 0 else 0)
            C = laser_amd.matmul(dA, dB).cpu().numpy()
            used = laser_amd.last_f32_asm() if dt == np.float32 else laser_amd.get_option("last_f64_asm")
            bad = int(np.sum(C != want))
            rec = {"dtype": np.dtype(dt).name, "scale": sc, "forced_kernel": kern, "kernel_used": used, "subnormal_results_frac": round(frac_sub, 4),
                   "mismatches": bad, "of": int(want.size)}
            if bad:
                i = np.argwhere(C != want)[0]
                rec["first"] = [float(C[tuple(i)]), float(want[tuple(i)])]
            print(json.dumps(rec), flush=True)
laser_amd.set_option("asm_kernel", -1); laser_amd.set_option("asm_plan", 0); laser_amd.set_option("f32_asm", 1)
