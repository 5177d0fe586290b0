#!/usr/bin/env python3
"""The reference's own conv bench shape, (16,3,224,224) (*) (20,3,3,3) pad 0 (conv2d_bench.nim:130-170), on the direct small-channel
kernels: option conv_direct = 1 (one pixel pair per lane, 8-byte stores) against 3 (two adjacent pairs per lane, 16-byte stores).
C-ABI symbol bound once; results compared bit for bit.
ONE-OFF history: commit 91f85e4 carried a 16-byte-store variant behind conv_direct = 3 (slower, removed:
profiles/r05/conv_small_quad_store_ab_v1.jsonl); a later one-off build carried a chunk loop (slower: conv_small_chunk_loop_ab_v1.jsonl) and
the channel-group split behind 3 / 4 (two groups kept as the default for one-round launches: conv_small_channel_groups_ab_v1.jsonl).
On the shipped library: conv_direct = 1 (default, with the split) against 3 (without it)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, laser_amd
from scripts.bench_configs import ev_time
L = laser_amd.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for ishape, kshape, pad in (((16, 3, 224, 224), (20, 3, 3, 3), (0, 0)), ((64, 3, 224, 224), (20, 3, 3, 3), (0, 0)), ((16, 3, 224, 224), (16, 3, 3, 3), (0, 0)),
                            ((16, 3, 226, 226), (20, 3, 3, 3), (0, 0)), ((16, 3, 224, 224), (20, 3, 3, 3), (1, 1)), ((16, 3, 225, 225), (24, 3, 3, 3), (0, 0)), ((8, 8, 112, 112), (24, 8, 3, 3), (1, 1))):
    x = torch.rand(ishape, device="cuda"); w = torch.rand(kshape, device="cuda")
    oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, (1, 1))
    outs = {}
    rec = {"conv": [ishape, kshape, pad]}
    for opt in (1, 3, 1, 3):
        laser_amd.set_option("conv_direct", opt)
        o = torch.zeros(oshape, device="cuda")
        args = (ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(x.data_ptr()), *ishape, ctypes.c_void_p(w.data_ptr()), *kshape, *pad, 1, 1, None, st)
        assert L.laser_hip_conv2d_im2col_f32_dev(*args) == 0
        med, mn = ev_time(lambda: L.laser_hip_conv2d_im2col_f32_dev(*args), iters=9, inner=16)
        outs[opt] = o
        byts = 4.0 * (x.numel() + o.numel())
        rec[f"conv_direct={opt}" + ("b" if f"conv_direct={opt}" in rec else "")] = {"us_med": round(med * 1e3, 2), "us_min": round(mn * 1e3, 2), "tbps": round(byts / (med * 1e-3) / 1e12, 2)}
    rec["bit_identical"] = bool(torch.equal(outs[1], outs[3]))
    laser_amd.set_option("conv_direct", 1)
    print(json.dumps(rec), flush=True)
