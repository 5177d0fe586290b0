#!/bin/bash
# round 6, the last call: the whole GPU suite, smoke and the driver's bench command on the committed tree, + the integer lines through the API
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-v12}
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_$T.log | tail -5
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; tail -1 $O/bench_$T.json | cut -c1-400
timeout 300 python scripts/int_gemm_ab.py 2>/dev/null > $O/int_gemm_ab_$T.jsonl; python - <<PY
import json
for l in open("$O/int_gemm_ab_$T.jsonl"):
    d = json.loads(l); print(d["dtype"], d["shape"][0], d["asm_ms"], d["asm_tintops"], "compiler", d["compiler_tintops"], d["bit_identical"])
PY
