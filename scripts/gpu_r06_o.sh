#!/bin/bash
# round 6, call O: the size sweep beside the vendor BLAS after the model's refit for one-chain cuts / strided launches / re-planned 16x16-block kernels; scheduler + parity tests that depend on the model's picks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-o}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/size_sweep_vendor.py 1024 8192 256 > $O/size_sweep_vendor_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/size_sweep_vendor_$T.jsonl"):
    d = json.loads(l); print(d["n"], d["laser_order_kernel"].replace("lh_", ""), d["laser_order_plan"], d["fast_kernel"].replace("lh_", ""), d["fast_plan"], "vendor", d["vendor_tflops"], "laser", d["laser_order_tflops"], "fast", d["fast_tflops"], "%+.1f %+.1f" % (d["laser_order_vs_vendor_pct"], d["fast_vs_vendor_pct"]))
PY
timeout 1500 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -x > $O/pytest_sched_parity_$T.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sched_parity_$T.log | cut -c1-300
