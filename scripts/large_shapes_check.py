#!/usr/bin/env python3
"""Large-extent sanity (GPU box): the C5 problem (65536 x 8192 x 8192) on ONE GPU, a very long K, a very wide N --
sampled rows bit-exact against the oracle, plus timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, laser_amd
from oracle import oracle
isa = oracle.fused_isa(np.float32)
def run(M, N, K, rows):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2
    B = (torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2
    C = torch.zeros((M, N), device="cuda")
    laser_amd.matmul(A, B, 1, 0, C); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); laser_amd.matmul(A, B, 1, 0, C); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    idx = sorted(set(rows))
    want = oracle.matmul(A[idx].cpu().numpy(), B.cpu().numpy(), isa=isa)
    ok = np.array_equal(C[idx].cpu().numpy(), want)
    print(f"{M}x{N}x{K}: {ms:.2f} ms {2.0*M*N*K/ms/1e9:.1f} TF, cfg {laser_amd.f32_configs()[laser_amd.last_f32_config()]}, "
          f"{len(idx)} sampled rows bit-exact: {ok}", flush=True)
    return ok
ok = run(65536, 8192, 8192, [0, 1, 255, 256, 32767, 32768, 65535, 40000])
ok &= run(512, 512, 131072, list(range(0, 512, 37)))
ok &= run(256, 262144, 512, [0, 100, 255])
ok &= run(1, 8192, 8192, [0])
ok &= run(8192, 1, 8192, list(range(0, 8192, 1000)))
sys.exit(0 if ok else 1)
