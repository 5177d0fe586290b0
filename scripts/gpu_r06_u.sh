#!/bin/bash
# round 6 (u): the one-latency prologue in the convolution kernels: parity first, then old/new builds alternated on the C4 shape
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/ -x -q -m gpu -k "conv" > $O/u_pytest_conv.log 2>&1; echo "pytest rc=$?" >> $O/u_pytest_conv.log
tail -3 $O/u_pytest_conv.log
timeout 600 python scripts/fuzz_conv.py 150 > $O/u_fuzz_conv.log 2>&1; tail -2 $O/u_fuzz_conv.log
for r in 1 2 3; do
  echo "== old $r" >> $O/u_c4_ab.log; timeout 300 python scripts/with_lib.py scripts/probes/ab_old/liblaser_hip.so scripts/conv_c4_run.py 20 >> $O/u_c4_ab.log 2>&1
  echo "== new $r" >> $O/u_c4_ab.log; timeout 300 python scripts/conv_c4_run.py 20 >> $O/u_c4_ab.log 2>&1
done
cat $O/u_c4_ab.log
echo "== old" > $O/u_conv_shapes_ab.log; timeout 300 python scripts/with_lib.py scripts/probes/ab_old/liblaser_hip.so scripts/conv_asm_ab.py >> $O/u_conv_shapes_ab.log 2>&1
echo "== new" >> $O/u_conv_shapes_ab.log; timeout 300 python scripts/conv_asm_ab.py >> $O/u_conv_shapes_ab.log 2>&1
