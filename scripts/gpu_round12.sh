#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python scripts/sweep_f32.py 8192,4096 4 0,1,4 > gpurun_out/sweep12.log 2>&1; echo "sweep rc=$?"; grep '"nn"' gpurun_out/sweep12.log | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
rows.sort(key=lambda r:(r['n'],-r['tflops_med']))
for r in rows: print(r['n'], r['cfg'], r['mode'], r['ms_med'], r['tflops_med'], r['frac_peak'])
"
bash scripts/gpu_profile_bench.sh default > gpurun_out/prof_default.log 2>&1; tail -30 gpurun_out/prof_default.log
