#!/bin/bash
# round 6, call A: the soffset bounds-check probe, the pipelined tile transitions A/B (plain vs strided persistent), parity of the scheduler tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-a}
O=gpurun_out/r06; mkdir -p $O
./scripts/probes/buffer_soffset > $O/buffer_soffset_$T.txt 2>&1; tail -4 $O/buffer_soffset_$T.txt
timeout 600 python scripts/pipe_ab.py big 3 > $O/pipe_ab_big_$T.jsonl 2> $O/pipe_ab_big_$T.err; cut -c1-600 $O/pipe_ab_big_$T.jsonl; tail -3 $O/pipe_ab_big_$T.err
timeout 900 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "scheduler or full_size or tile_config or asm" > $O/pytest_a_$T.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_a_$T.log
