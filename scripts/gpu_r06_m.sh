#!/bin/bash
# round 6, call M: what is left of the slice fold (ablations), subnormal operands through every tile family, the 16x16-block tiles with the running sum in VGPRs at their shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-m}
O=gpurun_out/r06; mkdir -p $O
for n in 8192 4096; do
timeout 900 python scripts/asm_probe.py scripts/asm_variants_fold.json --n $n --out $O/asm_probe_fold_n${n}_$T.jsonl > /dev/null 2> $O/asm_probe_fold_$T.err; python - <<PY
import json
print("n = $n")
for l in open("$O/asm_probe_fold_n${n}_$T.jsonl"):
    d = json.loads(l); print("  %-28s ms %.4f min %.4f frac %.4f err %s" % (d["variant"], d["ms_median"], d["ms_min"], d["frac_mfma_peak"], d["max_rel_err_vs_torch"]))
PY
done
timeout 600 python scripts/denormal_probe.py > $O/denormal_probe_$T.jsonl 2> $O/denormal_probe_$T.err; cut -c1-300 $O/denormal_probe_$T.jsonl; tail -2 $O/denormal_probe_$T.err
timeout 900 python scripts/x16_ab.py ref 3 > $O/x16_ab_ref_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/x16_ab_ref_$T.jsonl"):
    d = json.loads(l)
    print(d["M"], d["mode"], "best:", d["best_forced"], d["kernels"][d["best_forced"]].get("frac"), "model:", d["kernels"].get("model", {}).get("frac"), d["kernels"].get("model", {}).get("kernel_index"))
PY
timeout 600 python scripts/size_sweep_vendor.py 1536 5120 512 > $O/size_sweep_vendor_$T.jsonl 2> /dev/null; cut -c1-400 $O/size_sweep_vendor_$T.jsonl
