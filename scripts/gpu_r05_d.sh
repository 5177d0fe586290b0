#!/bin/bash
# round 5, call D: the direct conv tail kernel -- parity test, then the A/B against the round-3 tail forms; im2col with the band rule
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-d}
O=gpurun_out/r05; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "conv or im2col" > $O/pytest_conv_$T.log 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_conv_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error|assert|differs" | tail -20
timeout 300 python scripts/conv_tail_ab.py > $O/conv_tail_ab_$T.jsonl 2> $O/conv_tail_ab_$T.err; cut -c1-1200 $O/conv_tail_ab_$T.jsonl; tail -3 $O/conv_tail_ab_$T.err
timeout 100 python scripts/im2col_probe.py > $O/im2col_probe_$T.jsonl 2> $O/im2col_probe_$T.err; cut -c1-260 $O/im2col_probe_$T.jsonl
