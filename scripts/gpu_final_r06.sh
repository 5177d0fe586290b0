#!/bin/bash
# round 6, final call: full validation (suite, smoke, headline profile + traffic, bench, C3 / C4 profiles, N > 1 line formats, config lines) + fuzz campaigns + the size sweep, on the final sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-v3}
bash scripts/gpu_full_validation_r06.sh $T
O=gpurun_out/r06
timeout 2400 python scripts/fuzz_gemm.py 2000 71 > $O/fuzz_gemm_$T.log 2>&1; echo "fuzz gemm rc=$?"; tail -2 $O/fuzz_gemm_$T.log | cut -c1-300
timeout 1800 python scripts/fuzz_conv.py 800 72 > $O/fuzz_conv_$T.log 2>&1; echo "fuzz conv rc=$?"; tail -2 $O/fuzz_conv_$T.log | cut -c1-300
timeout 900 python scripts/size_sweep_vendor.py 1024 8192 256 > $O/size_sweep_vendor_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/size_sweep_vendor_$T.jsonl"):
    d = json.loads(l); print(d["n"], d["laser_order_kernel"].replace("lh_", ""), d["laser_order_plan"], "vendor", d["vendor_tflops"], "laser", d["laser_order_tflops"], "fast", d["fast_tflops"], "%+.1f %+.1f" % (d["laser_order_vs_vendor_pct"], d["fast_vs_vendor_pct"]))
PY
timeout 600 python scripts/vendor_blas_probe.py > $O/vendor_blas_large_$T.jsonl 2> /dev/null; cut -c1-330 $O/vendor_blas_large_$T.jsonl
timeout 600 python scripts/vendor_blas_probe.py small > $O/vendor_blas_small_$T.jsonl 2> /dev/null; cut -c1-330 $O/vendor_blas_small_$T.jsonl
