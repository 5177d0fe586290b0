#!/usr/bin/env python3
"""Build a VARIANT of liblaser_hip.so in which some hand-scheduled f32 kernels are generated with other schedule knobs, for old / new
A/B runs of the shipped launch paths on one box (scripts/with_lib.py <that library> <script>).  The product build is not touched:
objects are copied to build/laser_hip_<tag>, the named kernels' assembly is regenerated there, and the Makefile links
scripts/probes/ab_<tag>/liblaser_hip.so (git-ignored, travels with gpurun).

usage: build_variant_lib.py <tag> '{"conv_exact_256x128x32_p": {"bar_gap": 111}, ...}'"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_amd.asmgen import f32_kernel as K  # noqa: E402

tag, over = sys.argv[1], json.loads(sys.argv[2])
src, dst = os.path.join(ROOT, "build", "laser_hip"), os.path.join(ROOT, "build", "laser_hip_" + tag)
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)
for name, kw in over.items():
    g = K.make(name, **kw)
    g.build()
    sym = "lh_f32_" + name
    open(os.path.join(dst, "asm", sym + ".s"), "w").write(K.kernel_text(g, sym))
    print("regenerated", sym, kw)
for f in ("f32_asm.hsaco", "f32_asm_blob.h"):
    os.remove(os.path.join(dst, "asm", f))
os.utime(os.path.join(dst, "asm", ".generated"))      # newer than the generators: make does not regenerate
out = os.path.join(ROOT, "scripts", "probes", "ab_" + tag)
os.makedirs(out, exist_ok=True)
subprocess.check_call(["make", "-j8", "BUILD=" + os.path.relpath(dst, os.path.join(ROOT, "laser_amd", "csrc")),
                       "OUT=" + os.path.relpath(os.path.join(out, "liblaser_hip.so"), os.path.join(ROOT, "laser_amd", "csrc"))],
                      cwd=os.path.join(ROOT, "laser_amd", "csrc"), stdout=subprocess.DEVNULL)
print(os.path.join(out, "liblaser_hip.so"))
