#!/bin/bash
# round 6, call P: pipelined tile transitions on the 16x16-block tiles (plain vs strided plan, interleaved, same bits); the parity test of the family; the size sweep with the strided plans in
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-p}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/pipe_ab.py x16 3 > $O/pipe_ab_x16_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/pipe_ab_x16_$T.jsonl"):
    d = json.loads(l)
    if "plain" in d: print(d["M"], d["N"], d["K"], d["mode"], d["kernel"], "plain", d["plain"]["ms"], d["plain"]["frac"], d["plain"]["wgs"], "pipe", d["pipe"]["ms"], d["pipe"]["frac"], d["pipe"]["wgs"], "gain %", d["gain_pct"], d["bit_identical"])
    else: print(d)
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scheduler.py -m gpu -q --timeout 900 -x > $O/pytest_sched_parity_$T.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sched_parity_$T.log | cut -c1-300
timeout 900 python scripts/size_sweep_vendor.py 2560 8192 512 > $O/size_sweep_vendor_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/size_sweep_vendor_$T.jsonl"):
    d = json.loads(l); print(d["n"], d["laser_order_kernel"].replace("lh_", ""), d["laser_order_plan"], d["fast_kernel"].replace("lh_", ""), d["fast_plan"], "vendor", d["vendor_tflops"], "laser", d["laser_order_tflops"], "fast", d["fast_tflops"], "%+.1f %+.1f" % (d["laser_order_vs_vendor_pct"], d["fast_vs_vendor_pct"]))
PY
timeout 900 python scripts/fuzz_gemm.py 500 73 > $O/fuzz_gemm_$T.log 2>&1; echo "fuzz gemm rc=$?"; tail -2 $O/fuzz_gemm_$T.log | cut -c1-300
