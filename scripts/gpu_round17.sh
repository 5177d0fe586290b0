#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash scripts/gpu_profile_bench.sh default > gpurun_out/prof_default.log 2>&1; grep -A4 "kernel stats" gpurun_out/prof_default/summary.md | head -8; grep "FETCH_SIZE\|WRITE_SIZE\|TCC_HIT\|TCC_MISS" gpurun_out/prof_default/summary.md
bash scripts/gpu_profile_bench.sh fast --mode fast > gpurun_out/prof_fast.log 2>&1; grep -A4 "kernel stats" gpurun_out/prof_fast/summary.md | head -8; grep "FETCH_SIZE\|WRITE_SIZE" gpurun_out/prof_fast/summary.md
timeout 600 python bench.py > gpurun_out/bench17.json 2> gpurun_out/bench17.err; echo "bench rc=$?"; tail -c 900 gpurun_out/bench17.json
