#!/usr/bin/env python3
"""Schedule sweep of the hand-scheduled int32 limb kernel (laser_amd/asmgen/i8_kernel.py): every variant is generated, assembled,
loaded as its own code object and timed on random digit planes (the packing pass is not part of the timing; results are not
checked here -- tests/test_gpu_parity.py does).   usage: i8_probe.py variants.json [--n 8192] [--data random|zeros|low]   (a variant may carry "group_m": the raster's tile rows per group, default 8)
--data: the digit planes' content -- random bytes (default; what full-range operands give), zeros, or only the lowest plane random (operands in
[-128, 127]): the int8 matrix instructions' clock depends on the operand bits they toggle"""
import ctypes as C
import json, os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from laser_amd.asmgen import i8_kernel as KI, f32_kernel as K  # noqa: E402
from laser_amd.asmgen import check as CHK  # noqa: E402
from scripts.asm_probe import CLANG, LLD, hip  # noqa: E402

args = sys.argv[1:]
n = int(args[args.index("--n") + 1]) if "--n" in args else 8192
variants = json.load(open([a for a in args if a.endswith(".json")][0]))
tmp = tempfile.mkdtemp()
npad, kt = (n + 127) // 128 * 128, (n + 31) // 32
data = args[args.index("--data") + 1] if "--data" in args else "random"
Ap = torch.randint(-128, 128, (4 * npad * kt * 32,), dtype=torch.int8, device="cuda")
Bp = torch.randint(-128, 128, (4 * npad * kt * 32,), dtype=torch.int8, device="cuda")
if data == "zeros":
    Ap.zero_(); Bp.zero_()
elif data == "low":      # blocks are [plane p][k half][row][16 bytes] of 16 KiB: planes 1 .. 3 of every block cleared
    Ap.view(-1, 4, 4096)[:, 1:].zero_(); Bp.view(-1, 4, 4096)[:, 1:].zero_()
Cm = torch.zeros((n, n), dtype=torch.int32, device="cuda")
tm = npad // 128
st = torch.cuda.current_stream().cuda_stream
built = []
for var in variants:
    g = KI.make(**var.get("over", {}))
    g.build()
    sp = os.path.join(tmp, var["name"] + ".s")
    open(sp, "w").write(K.kernel_text(g, "lh_probe"))
    subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", sp, "-o", sp + ".o"])
    subprocess.check_call([LLD, "-shared", sp + ".o", "-o", sp + ".hsaco"])
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoad(C.byref(mod), (sp + ".hsaco").encode()) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, b"lh_probe") == 0
    # the kernel arguments as gemm_f32_asm.cpp: launch_gemm_i32_asm fills them (alpha = 1, beta = 0 as int32; the scheduler block of
    # the plain plan: one tile per workgroup, XCD remap, raster groups of 8 tile rows)
    ka = struct.pack("<QQQQIIIIIIiiQ", Ap.data_ptr(), Bp.data_ptr(), Cm.data_ptr(), 0, kt, 0, n, n, n, kt * 32, 1, 0, 0) + b"\0" * 80
    ka += CHK.sched_bytes(tm, tm, tm * tm, group_m=min(var.get("group_m", 8), tm), xcd=bool(var.get("xcd", True)))
    assert len(ka) == K.KERNARG_SIZE
    buf = C.create_string_buffer(ka, len(ka)); size = C.c_size_t(len(ka))
    extra = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
    built.append((var, fn, buf, size, extra))

def launch(b):
    rc = hip.hipModuleLaunchKernel(b[1], tm * tm, 1, 1, 256, 1, 1, 0, st, None, b[4])
    assert rc == 0, rc
for _ in range(8):
    launch(built[0])
torch.cuda.synchronize()
# every variant's C against the first variant's (the shipped kernel, which tests/test_gpu_parity.py holds to the oracle) on the same planes
same = {}
ref = None
for b in built:
    Cm.fill_(-7)
    launch(b); torch.cuda.synchronize()
    if ref is None:
        ref = Cm.clone()
    same[b[0]["name"]] = bool(torch.equal(Cm, ref))
res = {b[0]["name"]: [] for b in built}
for r in range(6):
    for b in built:
        launch(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            launch(b)
        e1.record(); torch.cuda.synchronize()
        if r:
            res[b[0]["name"]].append(e0.elapsed_time(e1) / 4)
for b in built:
    v_ = sorted(res[b[0]["name"]]); med = v_[len(v_) // 2]
    print(json.dumps({"variant": b[0]["name"], "over": b[0].get("over", {}), "n": n, "data": data, "same_c_as_first": same[b[0]["name"]], "ms_median": round(med, 4), "tintops": round(2.0 * n ** 3 / med / 1e9, 1),
                      "i8_tops": round(20.0 * n ** 3 / med / 1e9, 0)}), flush=True)
