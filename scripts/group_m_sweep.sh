#!/bin/bash
# fabric traffic of the headline launch (8192^3, laser-order 256x128 tile) by raster group height: FETCH_SIZE pass + timing per value
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04/group_m_sweep; rm -rf $O; mkdir -p $O
for gm in 1 2 4 8 16; do
  cat > /tmp/gm_run.py <<PY
import sys; sys.path.insert(0, "$PWD")
import laser_amd
laser_amd.set_option("asm_group_m", $gm)
sys.argv = ["shape_run.py", "8192", "8192", "8192", "0", "0", "1", "12"]
exec(open("$PWD/scripts/shape_run.py").read())
PY
  python /tmp/gm_run.py > $O/time_gm$gm.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_gm$gm -- python /tmp/gm_run.py > $O/fetch_gm$gm.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_gm$gm -- python /tmp/gm_run.py > $O/tcc_gm$gm.log 2>&1
done
python - <<PY
import csv, glob, json, os
from collections import defaultdict
out = []
for gm in (1, 2, 4, 8, 16):
    rec = {"group_m": gm}
    try:
        rec.update({k: v for k, v in json.load(open("$O/time_gm%d.json" % gm)).items() if k in ("ms", "tflops")})
    except Exception as e:
        rec["time_error"] = str(e)[:80]
    for tag in ("fetch", "tcc"):
        for f in glob.glob("$O/%s_gm%d/**/*_counter_collection.csv" % (tag, gm), recursive=True):
            per = defaultdict(lambda: defaultdict(float))
            for r in csv.DictReader(open(f)):
                if "lh_f32_exact_256x128x32" in r["Kernel_Name"]:
                    per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for c, d in per.items():
                rec[c] = sum(d.values()) / len(d)
            os.remove(f)
    if "FETCH_SIZE" in rec:
        rec["fetch_bytes_corrected"] = int(rec["FETCH_SIZE"] * 1024 * 2)      # KiB, doubled (gfx950: 128-B requests tallied at 64 B)
    if "TCC_HIT_sum" in rec:
        rec["tcc_hit_rate"] = round(rec["TCC_HIT_sum"] / (rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"]), 4)
    out.append(rec)
    print(json.dumps(rec))
open("$O/../group_m_sweep_v1.jsonl", "w").write("\n".join(json.dumps(r) for r in out) + "\n")
PY
rm -rf $O
