#!/bin/bash
# round 6 (ao): the integer limb kernels with LDS-DMA operand staging: parity (the integer GPU tests), then the library before / after
# alternated on one box: int32 / int64 n^3 through the API (packing pass included), bit identity against the compiler-scheduled limb kernel per line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "int or i32 or i64 or limb or integer" > $O/ao_pytest.log 2>&1; echo "pytest rc=$?" >> $O/ao_pytest.log
tail -3 $O/ao_pytest.log
: > $O/ao_int_dma_ab.log
for r in 1 2; do
  echo "== old $r" >> $O/ao_int_dma_ab.log; timeout 300 python scripts/with_lib.py scripts/probes/ab_old/liblaser_hip.so scripts/int_gemm_ab.py 2>/dev/null >> $O/ao_int_dma_ab.log
  echo "== new $r" >> $O/ao_int_dma_ab.log; timeout 300 python scripts/int_gemm_ab.py 2>/dev/null >> $O/ao_int_dma_ab.log
done
python - <<PY
import json
for l in open("$O/ao_int_dma_ab.log"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l); print(d["dtype"], d["shape"][0], d["asm_ms"], d["asm_tintops"], "compiler", d["compiler_tintops"], d["bit_identical"], d["asm_used_asm"])
PY
