#!/bin/bash
# round 6, call Q: the size sweep after the last model touch (class order, many-round efficiencies); the tile family's parity test with the strided plan forced
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-q}
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scheduler.py tests/test_gpu_sharded.py -m gpu -q --timeout 900 > $O/pytest_q_$T.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_q_$T.log | cut -c1-300
timeout 900 python scripts/size_sweep_vendor.py 1024 8192 256 > $O/size_sweep_vendor_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/size_sweep_vendor_$T.jsonl"):
    d = json.loads(l); print(d["n"], d["laser_order_kernel"].replace("lh_", ""), d["laser_order_plan"], d["fast_kernel"].replace("lh_", ""), d["fast_plan"], "vendor", d["vendor_tflops"], "laser", d["laser_order_tflops"], "fast", d["fast_tflops"], "%+.1f %+.1f" % (d["laser_order_vs_vendor_pct"], d["fast_vs_vendor_pct"]))
PY
timeout 900 python scripts/bench_configs.py > $O/configs_$T.jsonl 2> /dev/null; grep "int32\|int64" $O/configs_$T.jsonl | grep MFMA | cut -c1-260
