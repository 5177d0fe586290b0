#!/bin/bash
# round 5, call E: conv tail kernel -- parity tests, phase stamps of the kernel alone, the A/B against the round-3 tail forms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-e}
O=gpurun_out/r05; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "conv or im2col" > $O/pytest_conv_$T.log 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_conv_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error|assert|differs" | tail -20
bash scripts/gpu_ct_probe.sh $T | cut -c1-1400
timeout 300 python scripts/conv_tail_ab.py c4 > $O/conv_tail_ab_$T.jsonl 2> $O/conv_tail_ab_$T.err; cut -c1-1200 $O/conv_tail_ab_$T.jsonl; tail -3 $O/conv_tail_ab_$T.err
