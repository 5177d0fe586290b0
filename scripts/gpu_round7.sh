#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu7.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu7.log
timeout 600 python scripts/sweep_f32.py 8192,4096 4 > gpurun_out/sweep7.log 2>&1; echo "sweep rc=$?"
cp gpurun_out/sweep_f32.json gpurun_out/sweep_f32_v5.json
grep '^{' gpurun_out/sweep7.log | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['n'], r['cfg'], r['mode'], r['layout'], r['ms_med'], r['tflops_med'], r['frac_peak'])
"
