#!/bin/bash
# round 6, call B: pipelined transitions on the fixed sources (scalar offset inside the range check): A/B, then the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-b}
O=gpurun_out/r06; mkdir -p $O
timeout 600 python scripts/pipe_ab.py big 3 > $O/pipe_ab_big_$T.jsonl 2> $O/pipe_ab_big_$T.err; cut -c1-420 $O/pipe_ab_big_$T.jsonl; tail -3 $O/pipe_ab_big_$T.err
timeout 600 python scripts/pipe_ab.py mid 3 0,8,30,31 > $O/pipe_ab_mid_$T.jsonl 2> $O/pipe_ab_mid_$T.err; cut -c1-420 $O/pipe_ab_mid_$T.jsonl; tail -3 $O/pipe_ab_mid_$T.err
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu_$T.log
