#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu9.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu9.log | cut -c1-300
timeout 1200 python scripts/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo "configs rc=$?"; grep "C4\|C2\|C3" gpurun_out/configs.jsonl | cut -c1-200
