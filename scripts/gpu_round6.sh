#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu6.log
timeout 1200 python scripts/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo "configs rc=$?"; grep -c . gpurun_out/configs.jsonl
grep "C4\|ragged" gpurun_out/configs.jsonl | cut -c1-220
