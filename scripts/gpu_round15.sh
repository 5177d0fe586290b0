#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu15.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu15.log | cut -c1-300
timeout 600 python scripts/heuristic_check.py 4095x4097x4099,4100x4100x4100,1000x3000x2000,1001x3001x2001,4097x4097x4097,8191x8191x8191 2>&1 | grep -v amdgpu.ids > gpurun_out/hc15.jsonl; python - <<PY
import json
for l in open("gpurun_out/hc15.jsonl"):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r["shape"], r["mode"][:5], "chosen", r["chosen"][:10], "best", r["best"][:10], r["auto_ms"], r["auto_tflops"], r["auto_over_best"])
PY
