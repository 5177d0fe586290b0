"""Write-only, copy and read streams of the small-channel convolution's output size (63 MB, conv2d_bench.nim:130-170 geometry), 4x that
and 1 GiB: what a pure store stream reaches on this GPU, for a constant (fill_ / zero_) and for values that differ from element to
element (arange; a broadcast row copied down the buffer) -- the floor under any kernel whose traffic is mostly output.  One JSON line each."""
import json, torch
def t(fn, inner=20, reps=7):
    for _ in range(10): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]
for n in (16 * 20 * 222 * 222, 64 * 20 * 222 * 222, 1 << 28):
    o = torch.empty(n, device="cuda"); src = torch.rand(n, device="cuda")
    row = torch.rand(4096, device="cuda")
    o2 = o[: n // 4096 * 4096].view(-1, 4096)
    for name, fn, byts in (("fill_ (constant)", lambda: o.fill_(1.5), 4.0 * n), ("zero_", lambda: o.zero_(), 4.0 * n),
                           ("arange (distinct values, write only)", lambda: torch.arange(0, n, out=o), 4.0 * n),
                           ("random row broadcast down the buffer (write only)", lambda: o2.copy_(row.expand_as(o2)), 4.0 * o2.numel()),
                           ("copy_ (read + write)", lambda: o.copy_(src), 8.0 * n),
                           ("mul (read + write)", lambda: torch.mul(src, 2.0, out=o), 8.0 * n), ("sum (read only)", lambda: src.sum(), 4.0 * n)):
        ms = t(fn)
        print(json.dumps({"elements": n, "op": name, "us": round(ms * 1e3, 1), "TBps": round(byts / ms / 1e9, 2)}), flush=True)
