#!/bin/bash
# round 6, call E: the any-geometry assembly convolution loader -- parity tests, conv fuzz, the new config lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-e}
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scheduler.py -m gpu -q --timeout 900 -k "conv or im2col or strided or hand_over" > $O/pytest_e_$T.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_e_$T.log | cut -c1-400
timeout 900 python scripts/fuzz_conv.py 400 61 > $O/fuzz_conv_$T.log 2>&1; echo "fuzz rc=$?"; tail -6 $O/fuzz_conv_$T.log | cut -c1-300
timeout 900 python scripts/bench_configs.py > $O/configs_$T.jsonl 2> $O/configs_$T.err; echo "configs rc=$?"; grep -i "conv" $O/configs_$T.jsonl | cut -c1-420; tail -3 $O/configs_$T.err
