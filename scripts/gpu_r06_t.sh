#!/bin/bash
# round 6, call T: bench.py after the 1920^3 side line moved to the bound C-ABI symbol; interleaved LDS stages on the one-round tiles (does the shorter address set-up show?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-t}
O=gpurun_out/r06; mkdir -p $O
timeout 600 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; python - <<PY
import json
d = json.loads(open("$O/bench_$T.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for c in d["configs"]: print("   ", c["config"][:70], c.get("ms"), c.get("frac_mfma_peak"), c.get("python_mirror_ms"), c.get("kernel"))
PY
for n in 2048 1024; do
timeout 600 python scripts/asm_probe.py scripts/asm_variants_il.json --n $n --out $O/asm_probe_il_n${n}_$T.jsonl > /dev/null 2> $O/asm_probe_il_$T.err; python - <<PY
import json
print("n = $n")
for l in open("$O/asm_probe_il_n${n}_$T.jsonl"):
    d = json.loads(l); print("  %-24s wgs %5d ms %.4f min %.4f frac %.4f err %s" % (d["variant"], d["workgroups"], d["ms_median"], d["ms_min"], d["frac_mfma_peak"], d["max_rel_err_vs_torch"]))
PY
done
