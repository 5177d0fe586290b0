#!/bin/bash
# round 6 (z): plain / strided / K-cut plans on shapes a fraction of a round above whole rounds; the model's choice beside them
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python scripts/pipe_ab.py frac 3 0,2,30,1,8 cut > $O/z_plan_ab_frac.jsonl 2> $O/z_plan.err
python - <<PY
import json
for l in open("$O/z_plan_ab_frac.jsonl"):
    d = json.loads(l)
    if "skipped" in d: continue
    print(d["M"], d["mode"][:5], d["kernel"], "plain", d["plain"]["tflops"], d["plain"]["wgs"], "| strided", d["pipe"]["tflops"], d["pipe"]["wgs"], "| cut", d.get("cut", {}).get("tflops"), d.get("cut", {}).get("wgs"), d.get("cut", {}).get("slices"), d["bit_identical"])
PY
tail -2 $O/z_plan.err
timeout 600 python scripts/size_sweep_vendor.py 6656 7424 256 > $O/z_size_sweep.jsonl 2>/dev/null; cut -c1-400 $O/z_size_sweep.jsonl
