#!/bin/bash
# round 6, call K: the register re-plan of the laser-order headline kernel (interleaved LDS stages; running sum in arch VGPRs, fragments and staging in AGPRs) against the shipped kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-k}
O=gpurun_out/r06; mkdir -p $O
for n in 8192 4096; do
  timeout 900 python scripts/asm_probe.py scripts/asm_variants_runv.json --n $n --out $O/asm_probe_runv_n${n}_$T.jsonl > /dev/null 2> $O/asm_probe_runv_$T.err; python - <<PY
import json
print("n = $n")
for l in open("$O/asm_probe_runv_n${n}_$T.jsonl"):
    d = json.loads(l); print("  %-24s wgs %5d ms %.4f min %.4f frac %.4f err %s" % (d["variant"], d["workgroups"], d["ms_median"], d["ms_min"], d["frac_mfma_peak"], d["max_rel_err_vs_torch"]))
PY
done
timeout 600 python scripts/asm_probe.py scripts/asm_variants_runv.json --shape 8192 3072 1152 --out $O/asm_probe_runv_k1152_$T.jsonl > /dev/null 2>> $O/asm_probe_runv_$T.err; python - <<PY
import json
print("8192 x 3072 x 1152")
for l in open("$O/asm_probe_runv_k1152_$T.jsonl"):
    d = json.loads(l); print("  %-24s wgs %5d ms %.4f min %.4f frac %.4f err %s" % (d["variant"], d["workgroups"], d["ms_median"], d["ms_min"], d["frac_mfma_peak"], d["max_rel_err_vs_torch"]))
PY
tail -3 $O/asm_probe_runv_$T.err
