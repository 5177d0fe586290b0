#!/usr/bin/env python3
"""Run a script of this directory against ANOTHER build of liblaser_hip.so (A/B of two source states on one box, in one gpurun call):
with_lib.py <path/to/liblaser_hip.so> <script.py> [args...].  The Python mirror loads lazily, so the path is swapped before the first call."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import laser_amd._lib as _l

assert _l._lib is None, "the library was loaded at import time: the swap would be ignored"
_l.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
