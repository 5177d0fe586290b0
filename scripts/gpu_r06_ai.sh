#!/bin/bash
# round 6 (ai): barrier position / staging start of the 256-row convolution kernels (never swept on their own: bar_gap 95 came from the GEMM
# kernel's sweep, and the w_step A/B showed the convolution loses 2.7 % when its staging is packed): variant libraries alternated with the
# shipped one on the C4 shape; a parity check of each variant against the shipped build's output first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
O=gpurun_out/r06
: > $O/ai_conv_sched_ab.log
for r in 1 2 3; do
  for v in shipped cb79 cb111 cb119 cb111w8; do
    echo "== $v $r" >> $O/ai_conv_sched_ab.log
    if [ $v = shipped ]; then timeout 200 python scripts/conv_c4_run.py 20 >> $O/ai_conv_sched_ab.log 2>&1
    else timeout 200 python scripts/with_lib.py scripts/probes/ab_$v/liblaser_hip.so scripts/conv_c4_run.py 20 >> $O/ai_conv_sched_ab.log 2>&1; fi
  done
done
grep -v "^Hostname\|^Librccl\|amdgpu.ids" $O/ai_conv_sched_ab.log | cut -c1-200
