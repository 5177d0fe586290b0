#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/pytest_gpu20.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu20.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 1200 python scripts/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo "configs rc=$?"
timeout 900 python scripts/heuristic_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/heuristic_check.jsonl; echo "hc rc=$?"
bash scripts/gpu_profile_bench.sh default > gpurun_out/prof_default.log 2>&1
bash scripts/gpu_profile_bench.sh fast --mode fast > gpurun_out/prof_fast.log 2>&1
timeout 600 python bench.py > gpurun_out/bench20.json 2> gpurun_out/bench20.err; echo "bench rc=$?"; tail -c 700 gpurun_out/bench20.json
