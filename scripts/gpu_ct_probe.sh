#!/bin/bash
# the conv pixel-tail kernel alone with phase stamps (scripts/probes/conv_tail_timing.hip); extra -D flags: variants of the probe build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-p}; shift
O=gpurun_out/r05; mkdir -p $O
: > $O/conv_tail_timing_$T.jsonl
for V in "" "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I laser_amd/csrc $V scripts/probes/conv_tail_timing.hip -o /tmp/ct_probe 2> /tmp/ct_build.log || { grep error /tmp/ct_build.log | head -5; continue; }
  echo "{\"variant\": \"$V\"}" >> $O/conv_tail_timing_$T.jsonl
  timeout 60 /tmp/ct_probe >> $O/conv_tail_timing_$T.jsonl 2>&1
  timeout 60 /tmp/ct_probe >> $O/conv_tail_timing_$T.jsonl 2>&1
done
cat $O/conv_tail_timing_$T.jsonl
