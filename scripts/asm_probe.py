#!/usr/bin/env python3
"""Schedule sweep of the hand-scheduled fp32 kernels (laser_amd/asmgen/f32_kernel.py): every variant is generated,
assembled and loaded as its own code object, launched under the plain plan (one tile per workgroup, the kernel's own tile
map), checked against torch.matmul (unless it is an ablation) and timed at 8192^3 (or --shape M N K) in interleaved rounds.  Writes one JSON line per variant (profiles/r03/asm_probe_*.jsonl are copies of that).

usage: asm_probe.py [variants.json] [--n 8192] [--out file.jsonl]
variants.json: [{"name": "...", "kernel": "exact_256x128x32", "over": {"bar_gap": 63}}, ...]"""
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from laser_amd.asmgen import f32_kernel as K  # noqa: E402
from laser_amd.asmgen import f32x16_kernel as K16  # noqa: E402
from laser_amd.asmgen import check as CHK  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang"
LLD = "/opt/rocm/lib/llvm/bin/ld.lld"
hip = C.CDLL("libamdhip64.so")
hip.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
hip.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]


def build(var, tmp):
    g = (K16 if var.get("module") == "x16" else K).make(var["kernel"], **var.get("over", {}))      # "module": "x16" = the 16x16-block family
    g.build()
    sym = "lh_probe_" + "".join(ch if ch.isalnum() else "_" for ch in var["name"])   # one symbol per variant: rocprofv3 rows stay apart
    spath = os.path.join(tmp, var["name"] + ".s")
    open(spath, "w").write(K.kernel_text(g, sym))
    subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", spath, "-o", spath + ".o"])
    subprocess.check_call([LLD, "-shared", spath + ".o", "-o", spath + ".hsaco"])
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoad(C.byref(mod), (spath + ".hsaco").encode()) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, sym.encode()) == 0
    return g.c, fn


def main():
    args = sys.argv[1:]
    n = 8192
    out = None
    if "--n" in args:
        n = int(args[args.index("--n") + 1])
    if "--out" in args:
        out = args[args.index("--out") + 1]
    shape = (n, n, n)
    if "--shape" in args:
        i = args.index("--shape")
        shape = tuple(int(x) for x in args[i + 1:i + 4])
    M_, N_, K_ = shape
    files = [a for a in args if a.endswith(".json")]
    variants = json.load(open(files[0])) if files else [
        {"name": "exact_base", "kernel": "exact_256x128x32"},
        {"name": "fast_base", "kernel": "fast_256x256x16"},
    ]
    torch.manual_seed(0)
    A = (torch.rand((M_, K_), device="cuda") - 0.5) * 0.2
    B = (torch.rand((K_, N_), device="cuda") - 0.5) * 0.2
    Cm = torch.zeros((M_, N_), device="cuda")
    ref = None
    st = torch.cuda.current_stream().cuda_stream
    tmp = tempfile.mkdtemp()
    built = []
    for var in variants:
        cfg, fn = build(var, tmp)
        tm, tn = (M_ + cfg.BM - 1) // cfg.BM, (N_ + cfg.BN - 1) // cfg.BN
        gm = 4 if cfg.BM >= 2 * cfg.BN else 8
        # the kernel arguments as gemm_f32_asm.cpp fills them for the plain plan: one tile per workgroup, XCD-aware remap, raster
        # groups of gm tile rows, tiles never cut (f32_kernel.py: KA_SCHED .. KERNARG_SIZE)
        ka = struct.pack("<QQQQIIIIIIffQ", A.data_ptr(), B.data_ptr(), Cm.data_ptr(), 0, K_, N_, N_, M_, N_, K_, 1.0, 0.0, 0) + b"\0" * 80
        assert len(ka) == K.KA_SCHED
        ka += CHK.sched_bytes(tm, tn, tm * tn, group_m=min(gm, tm), xcd=True)
        assert len(ka) == K.KERNARG_SIZE, len(ka)
        buf = C.create_string_buffer(ka, len(ka))
        size = C.c_size_t(len(ka))
        extra = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
        built.append((var, cfg, fn, tm * tn, buf, size, extra))

    def launch(b):
        rc = hip.hipModuleLaunchKernel(b[2], b[3], 1, 1, 256, 1, 1, 0, st, None, b[6])
        assert rc == 0, rc

    res = {b[0]["name"]: [] for b in built}
    err = {}
    for b in built:
        Cm.zero_()
        launch(b)
        torch.cuda.synchronize()
        if not b[0].get("over", {}).get("ablate"):
            if ref is None:
                ref = A @ B
            err[b[0]["name"]] = float(((Cm - ref).abs().max() / ref.abs().max()).item())
    for _ in range(max(1, int(0.03 / max(1e-6, 2.0 * M_ * N_ * K_ / 100e12)))):     # clocks up: ~30 ms of work
        launch(built[0])
    torch.cuda.synchronize()
    for r in range(6):
        for b in built:
            launch(b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                launch(b)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[b[0]["name"]].append(e0.elapsed_time(e1) / 4)
    lines = []
    for b in built:
        v_ = sorted(res[b[0]["name"]])
        med = v_[len(v_) // 2]
        fl = 2.0 * M_ * N_ * K_
        line = {"variant": b[0]["name"], "kernel": b[0]["kernel"], "over": b[0].get("over", {}), "shape": [M_, N_, K_], "workgroups": b[3], "ms_median": round(med, 4),
                "ms_min": round(v_[0], 4), "tflops": round(fl / med / 1e9, 1), "frac_mfma_peak": round(fl / med / 1e9 / 157.3, 4),
                "max_rel_err_vs_torch": err.get(b[0]["name"])}
        print(json.dumps(line), flush=True)
        lines.append(line)
    if out:
        with open(out, "w") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
