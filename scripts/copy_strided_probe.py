#!/usr/bin/env python3
"""deepCopy / copyFrom throughput on typical views (GPU box) and equality with numpy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, laser_amd as la
def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
base = la.fromTorch(torch.rand((8192, 8192), device="cuda"))
ref = torch.as_tensor(base, device="cuda")
cases = {"row slice [1000:7000, :]": (base[1000:7000, :], ref[1000:7000, :]),
         "column slice [:, 100:6244]": (base[:, 100:6244], ref[:, 100:6244]),
         "every 2nd row [::2, :]": (base[::2, :], ref[::2, :]),
         "every 2nd column [:, ::2]": (base[:, ::2], ref[:, ::2]),
         "transpose .T": (base.T, ref.t()),
         "reversed rows [::-1, :]": (base[::-1, :], ref.flip(0)),
         "4-D permute (64,128,64,128)->(0,2,1,3)": None}
t4 = la.fromTorch(torch.rand((64, 128, 64, 128), device="cuda")); r4 = torch.as_tensor(t4, device="cuda")
cases["4-D permute (64,128,64,128)->(0,2,1,3)"] = (t4.transpose(0, 2, 1, 3), r4.permute(0, 2, 1, 3))
for name, (v, rv) in cases.items():
    out = la.newTensor(np.float32, *v.shape)          # copyFrom into existing storage (deepCopy = this + an allocation)
    la.copyFrom(out, v)
    ok = torch.equal(torch.as_tensor(out, device="cuda"), rv.contiguous())
    ms = bench(lambda: la.copyFrom(out, v))
    gb = 2 * v.size * 4 / 1e9
    print(f"{name:42s} {ms:.3f} ms  {gb/ms:6.2f} TB/s  {'ok' if ok else 'WRONG'}")
