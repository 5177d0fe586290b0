#!/bin/bash
# round 6 (ab): the hybrid plan (strided whole rounds + a K-cut launch over the remaining tiles): parity tests, forced A/B, the size sweep
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_parity.py -x -q -m gpu -k "plan or sched or mid_sizes or 16x16 or block_tiles" > $O/ab_pytest.log 2>&1; echo "pytest rc=$?" >> $O/ab_pytest.log
tail -4 $O/ab_pytest.log
timeout 900 python scripts/pipe_ab.py frac 3 0,30,1,8 cut hyb > $O/ab_plan_ab_frac.jsonl 2> $O/ab_plan.err
python - <<PY
import json
for l in open("$O/ab_plan_ab_frac.jsonl"):
    d = json.loads(l)
    if "skipped" in d: continue
    h = d.get("hybrid", {})
    print(d["M"], d["mode"][:5], d["kernel"], "plain", d["plain"]["tflops"], "| strided", d["pipe"]["tflops"], "| cut", d.get("cut", {}).get("tflops"), "| hybrid", h.get("tflops"), h.get("wgs"), h.get("slices"), h.get("rem_tiles"), d["bit_identical"])
PY
tail -2 $O/ab_plan.err
timeout 900 python scripts/size_sweep_vendor.py 4096 8192 256 > $O/ab_size_sweep.jsonl 2>/dev/null
python - <<PY
import json
for l in open("$O/ab_size_sweep.jsonl"):
    d = json.loads(l); print(d["n"], d["laser_order_kernel"].replace("lh_", ""), d["laser_order_plan"], "vendor", d["vendor_tflops"], "laser", d["laser_order_tflops"], "fast", d["fast_tflops"], "%+.1f %+.1f" % (d["laser_order_vs_vendor_pct"], d["fast_vs_vendor_pct"]))
PY
