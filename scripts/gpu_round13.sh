#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu13.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu13.log | cut -c1-300
timeout 900 python scripts/sweep_f32.py 8192,4096 4 0,1,2,4 > gpurun_out/sweep13.log 2>&1; echo "sweep rc=$?"; grep '"layout"' gpurun_out/sweep13.log | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
rows.sort(key=lambda r:(r['n'],r['layout'],-r['tflops_med']))
for r in rows: print(r['n'], r['layout'], r['cfg'], r['mode'], r['ms_med'], r['tflops_med'], r['frac_peak'])
"
