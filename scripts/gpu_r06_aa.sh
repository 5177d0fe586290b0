#!/bin/bash
# round 6 (aa): float64 pipelined transitions: GPU tests, A/B plain vs strided vs the model's choice
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_parity.py -x -q -m gpu -k "f64 or float64 or 16x16 or plans or pipelined" > $O/aa_pytest.log 2>&1; echo "pytest rc=$?" >> $O/aa_pytest.log
tail -4 $O/aa_pytest.log
timeout 900 python scripts/f64_pipe_ab.py 1536,2048,3072,4096,6144,8192 3 > $O/aa_f64_pipe_ab.jsonl 2> $O/aa_f64.err
python - <<PY
import json
for l in open("$O/aa_f64_pipe_ab.jsonl"):
    d = json.loads(l)
    print(d["n"], d["mode"][:5], "plain", d["plain"]["tflops"], d["plain"]["kernel"], d["plain"]["wgs"], "| strided", d["strided"]["tflops"], d["strided"]["frac"], d["strided"]["wgs"], "| model", d["model"]["tflops"], d["model"]["frac"], d["model"]["kernel"], d["model"]["wgs"], d["model"]["slices"], d["gain_pct"], d["plain_eq_strided"], d["model_eq_plain"])
PY
tail -2 $O/aa_f64.err
