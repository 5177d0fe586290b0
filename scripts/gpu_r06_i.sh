#!/bin/bash
# round 6, call I: the whole GPU suite on the widened 16x16-block family (128x96, 192x96, 160x160 added), A/B at the shapes the new tiles were built for
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-i}
O=gpurun_out/r06; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed" | tail -12
timeout 1500 python scripts/x16_ab.py more 3 > $O/x16_ab_more_$T.jsonl 2> $O/x16_ab_more_$T.err; python - <<PY
import json
for l in open("$O/x16_ab_more_$T.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["mode"], "best:", d["best_forced"], d["kernels"][d["best_forced"]].get("frac"), "model:", d["kernels"].get("model", {}).get("frac"), d["kernels"].get("model", {}).get("kernel_index"), "model vs best %:", d["model_vs_best_pct"])
    for k, v in d["kernels"].items():
        if "16x16" in k and "plain" in k and "ms" in v: print("      ", k, v["ms"], v["frac"], v.get("wgs"))
PY
tail -3 $O/x16_ab_more_$T.err
