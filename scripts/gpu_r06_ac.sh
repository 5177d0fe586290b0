#!/bin/bash
# round 6 (ac): hybrid plan, one-chain mode only in the model: scheduler tests, fuzz with plan 4 in the draw, the size sweep
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_scheduler.py -x -q -m gpu > $O/ac_pytest.log 2>&1; echo "pytest rc=$?" >> $O/ac_pytest.log
tail -4 $O/ac_pytest.log
timeout 900 python scripts/fuzz_gemm.py 500 81 > $O/ac_fuzz_gemm.log 2>&1; tail -2 $O/ac_fuzz_gemm.log | cut -c1-300
timeout 1200 python scripts/size_sweep_vendor.py 1024 8192 256 > $O/ac_size_sweep.jsonl 2>/dev/null
python - <<PY
import json
for l in open("$O/ac_size_sweep.jsonl"):
    d = json.loads(l); print(d["n"], d["laser_order_kernel"].replace("lh_", ""), d["laser_order_plan"], d["fast_kernel"].replace("lh_", ""), d["fast_plan"], "vendor", d["vendor_tflops"], "laser", d["laser_order_tflops"], "fast", d["fast_tflops"], "%+.1f %+.1f" % (d["laser_order_vs_vendor_pct"], d["fast_vs_vendor_pct"]))
PY
