#!/bin/bash
# round 6, call J: the launch model after the refit (model vs best forced, mid shapes), square sizes beside the vendor BLAS, GEMM fuzz with the new tiles in the draw, counters of the small-channel convolution
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-j}
O=gpurun_out/r06; mkdir -p $O
for L in mid more; do
timeout 1500 python scripts/x16_ab.py $L 3 > $O/x16_ab_${L}_$T.jsonl 2> $O/x16_ab_${L}_$T.err; python - <<PY
import json
for l in open("$O/x16_ab_${L}_$T.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["mode"], "best:", d["best_forced"], d["kernels"][d["best_forced"]].get("frac"), "model:", d["kernels"].get("model", {}).get("frac"), d["kernels"].get("model", {}).get("kernel_index"), "model vs best %:", d["model_vs_best_pct"])
PY
done
timeout 900 python scripts/size_sweep_vendor.py 1024 6144 256 > $O/size_sweep_vendor_$T.jsonl 2> $O/size_sweep_vendor_$T.err; cut -c1-400 $O/size_sweep_vendor_$T.jsonl; tail -2 $O/size_sweep_vendor_$T.err
timeout 1200 python scripts/fuzz_gemm.py 700 66 > $O/fuzz_gemm_$T.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz_gemm_$T.log | cut -c1-300
SKIP=20 timeout 900 bash scripts/gpu_profile_cmd.sh conv_small python scripts/probes/conv_direct_loop.py 1 16 > /dev/null 2>&1; rm -rf $O/rocprof_conv_small; cp -r gpurun_out/prof_conv_small $O/rocprof_conv_small; grep -v "^$" $O/rocprof_conv_small/summary.md | head -60 | cut -c1-200
