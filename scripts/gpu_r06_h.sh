#!/bin/bash
# round 6, call H: where the 16x16-block tiles lose their time (ablations, barrier position), the vendor yardstick at the small shapes with the new tiles, sharded / scheduler tests after the thread pin
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-h}
O=gpurun_out/r06; mkdir -p $O
for n in 1536 1920 3072; do
  timeout 600 python scripts/asm_probe.py scripts/asm_variants_x16.json --n $n --out $O/asm_probe_x16_n${n}_$T.jsonl > /dev/null 2> $O/asm_probe_x16_$T.err; python - <<PY
import json
print("n = $n")
for l in open("$O/asm_probe_x16_n${n}_$T.jsonl"):
    d = json.loads(l); print("  %-24s wgs %5d ms %.4f frac %.4f err %s" % (d["variant"], d["workgroups"], d["ms_median"], d["frac_mfma_peak"], d["max_rel_err_vs_torch"]))
PY
done
tail -2 $O/asm_probe_x16_$T.err
timeout 600 python scripts/vendor_blas_probe.py small > $O/vendor_blas_small_v2.jsonl 2> $O/vendor_blas_small_v2.err; cut -c1-330 $O/vendor_blas_small_v2.jsonl
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_scheduler.py -m gpu -q --timeout 900 -x > $O/pytest_shard_$T.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_shard_$T.log | cut -c1-400
