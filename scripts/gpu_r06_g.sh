#!/bin/bash
# round 6, call G: the 16x16-block tile family (96x96, 160x96) -- parity test, A/B against the 32x32-block tiles at the mid-size shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-g}
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -k "16x16_block or f32_asm_kernels_bit_exact or any_matrix_view" > $O/pytest_x16_$T.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_x16_$T.log | cut -c1-600
timeout 900 python scripts/x16_ab.py ref 3 > $O/x16_ab_ref_$T.jsonl 2> $O/x16_ab_ref_$T.err; python - <<PY
import json
for l in open("$O/x16_ab_ref_$T.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["mode"], "best:", d["best_forced"], "model vs best %:", d["model_vs_best_pct"])
    for k, v in d["kernels"].items():
        print("   ", k, {x: v.get(x) for x in ("ms", "frac", "wgs", "slices", "kernel_index", "same_bits_as_first", "skipped") if x in v})
PY
tail -3 $O/x16_ab_ref_$T.err
timeout 1200 python scripts/x16_ab.py mid 3 > $O/x16_ab_mid_$T.jsonl 2> $O/x16_ab_mid_$T.err; python - <<PY
import json
for l in open("$O/x16_ab_mid_$T.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["mode"], "best:", d["best_forced"], d["kernels"][d["best_forced"]].get("frac"), "model:", d["kernels"].get("model", {}).get("frac"), d["kernels"].get("model", {}).get("kernel_index"), "model vs best %:", d["model_vs_best_pct"])
PY
tail -3 $O/x16_ab_mid_$T.err
