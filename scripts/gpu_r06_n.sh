#!/bin/bash
# round 6, call N: the launch model at the large sizes where the sweep shows it behind the vendor BLAS (5632, 6912, 7936, 3328): every candidate forced
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-n}
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python scripts/x16_ab.py big2 2 > $O/x16_ab_big2_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/x16_ab_big2_$T.jsonl"):
    d = json.loads(l)
    print(d["M"], d["mode"], "best:", d["best_forced"], d["kernels"][d["best_forced"]].get("frac"), "model:", d["kernels"].get("model", {}).get("frac"), d["kernels"].get("model", {}).get("kernel_index"), d["kernels"].get("model", {}).get("wgs"), d["kernels"].get("model", {}).get("slices"))
    for k, v in sorted(d["kernels"].items(), key=lambda kv: kv[1].get("ms", 9e9))[:6]:
        if "ms" in v: print("      %-42s ms %.4f frac %.4f wgs %s slices %s" % (k, v["ms"], v["frac"], v.get("wgs"), v.get("slices")))
PY
