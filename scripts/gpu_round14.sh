#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu14.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu14.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench14.json 2> gpurun_out/bench14.err; echo "bench rc=$?"; cat gpurun_out/bench14.json | cut -c1-1200
timeout 1200 python scripts/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo "configs rc=$?"; grep "C2\|C3\|C4 conv.*implicit\|ragged" gpurun_out/configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['config'][:64], r.get('mode',''), r.get('ms_med'), r.get('tflops',''))"
