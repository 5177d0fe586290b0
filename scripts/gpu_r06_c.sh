#!/bin/bash
# round 6, call C: the whole GPU suite on the pipelined-transition sources + the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-c}
O=gpurun_out/r06; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu_$T.log | cut -c1-300
timeout 600 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; echo "bench rc=$?"; cut -c1-1500 $O/bench_$T.json; tail -3 $O/bench_$T.err
