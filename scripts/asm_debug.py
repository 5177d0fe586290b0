#!/usr/bin/env python3
"""Debug build of a hand-scheduled kernel (Cfg(debug=True)): workgroup 0 dumps intermediate registers / LDS words to a
buffer; the same instruction list runs in the CPU interpreter and (when a GPU is present) on the hardware, and the two
dumps are compared slot by slot.  usage: asm_debug.py [kernel] [M N K]"""
import ctypes as C
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_amd.asmgen import f32_kernel as K  # noqa: E402
from laser_amd.asmgen.sim import Memory, Workgroup  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "exact_256x128x32"
g = K.make(name, debug=True)
g.build()
c = g.c
rng = np.random.default_rng(0)
NIMG = 1
if c.conv:   # usage: asm_debug.py conv_... images Cin H W M pad
    NIMG, Cin, H, W, M, pad = (int(x) for x in sys.argv[2:8]) if len(sys.argv) > 7 else (2, 8, 12, 16, 40, 1)
    oH, oW = H + 2 * pad - 2, W + 2 * pad - 2
    N, Kd = oH * oW, Cin * 9
    A = rng.uniform(-0.1, 0.1, (M, Kd)).astype(np.float32)
    B = rng.uniform(-0.1, 0.1, (NIMG, Cin, H, W)).astype(np.float32)
    xp = np.zeros((NIMG, Cin, H + 2 * pad, W + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = B
    want = np.stack([A.astype(np.float64) @ np.stack([xp[b, ci, kh:kh + oH, kw:kw + oW].reshape(-1) for ci in range(Cin) for kh in range(3) for kw in range(3)])
                     for b in range(NIMG)])
    conv_args = struct.pack("<IIIIIIII", H, W, oW, pad, pad, Cin, N, (1 << 32) // oW + 1) + struct.pack("<IIQ", 0, 0, Cin * H * W * 4) + struct.pack("<Q", M * N * 4) + b"\0" * 24
    lds = (Kd, 0, N)
else:
    M, N, Kd = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (c.BM, c.BN, 2 * c.BK)
    A = rng.uniform(-0.1, 0.1, (M, Kd)).astype(np.float32)
    B = rng.uniform(-0.1, 0.1, (Kd, N)).astype(np.float32)
    want = (A.astype(np.float64) @ B.astype(np.float64))[None]
    conv_args = b"\0" * 80
    lds = (Kd, N, N)
tm, tn = (M + c.BM - 1) // c.BM, (N + c.BN - 1) // c.BN
table = np.array([pm | (pn << 16) for pn in range(tn) for pm in range(tm)], dtype=np.uint32)
NS = g.ndump

# ---- interpreter ----
mem = Memory()
a_, b_, c_, t_ = mem.alloc(A), mem.alloc(B), mem.alloc(np.full((NIMG, M, N), np.nan, np.float32)), mem.alloc(table)
d_ = mem.alloc(np.zeros(NS * 256 + 64, dtype=np.uint32))
ka_ = mem.alloc(np.frombuffer(struct.pack("<QQQQIIIIIIffQ", a_, b_, c_, t_, lds[0], lds[1], lds[2], M, N, Kd, 1.0, 0.0, d_) + conv_args, dtype=np.uint8))
for img in range(NIMG):
    for wg in range(len(table)):
        Workgroup(g.p, mem, ka_, wg_id=(wg, img), lds_bytes=c.lds_alloc).run()
sim_dump = mem.get(d_, np.uint32, (NS * 256 + 64,))[:NS * 256].reshape(NS, 256).copy()
sim_C = mem.get(c_, np.float32, (NIMG, M, N)).copy()
print("interpreter: C max abs err vs fp64", float(np.nanmax(np.abs(sim_C - want))), "nan count", int(np.isnan(sim_C).sum()))

# ---- hardware ----
import torch  # noqa: E402
if not torch.cuda.is_available():
    print("no GPU: interpreter only;", NS, "dump slots")
    for i, nm in enumerate(g.dump_names):
        print(f"  slot {i:3d} {nm:16s} lane0..3 = {[hex(int(x)) for x in sim_dump[i][:4]]}")
    sys.exit(0)
tmp = tempfile.mkdtemp()
sp = os.path.join(tmp, "k.s")
open(sp, "w").write(K.kernel_text(g, "lh_dbg"))
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", sp, "-o", sp + ".o"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/ld.lld", "-shared", sp + ".o", "-o", sp + ".hsaco"])
hip = C.CDLL("libamdhip64.so")
hip.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
hip.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
mod, fn = C.c_void_p(), C.c_void_p()
assert hip.hipModuleLoad(C.byref(mod), (sp + ".hsaco").encode()) == 0
assert hip.hipModuleGetFunction(C.byref(fn), mod, b"lh_dbg") == 0
dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
dC = torch.full((NIMG, M, N), float("nan"), device="cuda")
dT = torch.from_numpy(table.astype(np.int32)).cuda()
dD = torch.zeros(NS * 256 + 64, dtype=torch.int32, device="cuda")
ka = struct.pack("<QQQQIIIIIIffQ", dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), dT.data_ptr(), lds[0], lds[1], lds[2], M, N, Kd, 1.0, 0.0, dD.data_ptr()) + conv_args
buf = C.create_string_buffer(ka, len(ka))
size = C.c_size_t(len(ka))
extra = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
rc = hip.hipModuleLaunchKernel(fn, len(table), NIMG, 1, 256, 1, 1, 0, torch.cuda.current_stream().cuda_stream, None, extra)
assert rc == 0, rc
torch.cuda.synchronize()
hw_dump = dD.cpu().numpy().view(np.uint32)[:NS * 256].reshape(NS, 256)
hw_C = dC.cpu().numpy()
print("hardware:    C max abs err vs fp64", float(np.nanmax(np.abs(hw_C - want))) if not np.all(np.isnan(hw_C)) else "all nan",
      "nan count", int(np.isnan(hw_C).sum()), "bit-identical to interpreter:", bool(np.array_equal(hw_C, sim_C, equal_nan=True)))
addr_like = ("srdA[0]", "srdA[1]", "srdB[0]", "srdB[1]", "srdC[0]", "srdC[1]")
for i, nm in enumerate(g.dump_names):
    same = np.array_equal(sim_dump[i], hw_dump[i])
    note = " (address: differs by construction)" if nm in addr_like else ""
    bad = np.flatnonzero(sim_dump[i] != hw_dump[i])
    print(f"  slot {i:3d} {nm:18s} {'same' if same else 'DIFF'}{note}" + ("" if same else
          f"  first bad lane {bad[0]} of {len(bad)}: sim {hex(int(sim_dump[i][bad[0]]))} hw {hex(int(hw_dump[i][bad[0]]))}; lane0 sim {hex(int(sim_dump[i][0]))} hw {hex(int(hw_dump[i][0]))}"))
