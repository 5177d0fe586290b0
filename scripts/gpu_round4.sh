#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu4.log
timeout 600 python scripts/sweep_f32.py 8192,4096,2048,1024 3 > gpurun_out/sweep4.log 2>&1; echo "sweep rc=$?"
cp gpurun_out/sweep_f32.json gpurun_out/sweep_f32_v4.json
grep '^{' gpurun_out/sweep4.log | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['n'], r['cfg'], r['mode'], r['layout'], r['ms_med'], r['tflops_med'], r['frac_peak'])
"
timeout 900 python scripts/bench_configs.py > gpurun_out/configs.log 2>&1; echo "configs rc=$?"; grep '^{' gpurun_out/configs.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
timeout 300 python bench.py --mode fast --no-cpu-baseline > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err; cat gpurun_out/bench_fast.json
