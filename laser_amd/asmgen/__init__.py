"""Generator (and CPU interpreter) of the hand-scheduled gfx950 assembly kernels of liblaser_hip.so.

The reference generates its register-blocked micro-kernels with a Nim macro that places every load, broadcast and FMA
(laser/primitives/matrix_multiplication/gemm_ukernel_generator.nim:140-250).  The MI355X twin is this package: a Python
program that emits the whole fp32 GEMM kernel as CDNA4 assembly -- every v_mfma, ds_read, ds_write, buffer_load and
counted s_waitcnt placed by hand -- and `sim.py`, an instruction-level interpreter of the emitted stream (registers,
LDS, wait counters, barrier epochs) that checks results, wait counts, LDS races and bank conflicts on the CPU before a
GPU ever runs it.  Build-time only: nothing here is imported by the product path at run time.
"""
