#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "tile_config or fast_mode or bit_exact_vs_oracle or full_size_8192" > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu2.log
timeout 900 python scripts/sweep_f32.py 8192 3 > gpurun_out/sweep2.log 2>&1; echo "sweep rc=$?"; grep '"nn"' gpurun_out/sweep2.log | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
rows.sort(key=lambda r:-r['tflops_med'])
for r in rows: print(r['cfg'], r['mode'], r['ms_med'], r['tflops_med'], r['frac_peak'])
"
