#!/bin/bash
# round 6, call L: the register re-plan (running sum in arch VGPRs; interleaved LDS stages) shipped on every laser-order kernel: whole GPU suite, bench line, config lines, fuzz
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-l}
O=gpurun_out/r06; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed" | tail -12
timeout 500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; tail -1 $O/bench_$T.json | cut -c1-900; python - <<PY
import json
d = json.loads(open("$O/bench_$T.json").read().strip().splitlines()[-1])
for c in d["configs"]: print("   ", c["config"][:70], c.get("ms"), c.get("frac_mfma_peak"), c.get("hbm_frac"))
PY
timeout 600 python scripts/pipe_ab.py head 3 > $O/pipe_ab_head_$T.jsonl 2> /dev/null; python - <<PY
import json
for l in open("$O/pipe_ab_head_$T.jsonl"):
    d = json.loads(l); print(d["M"], d["mode"], d["kernel"], "plain", d.get("plain", {}).get("ms"), d.get("plain", {}).get("frac"), "pipe", d.get("pipe", {}).get("ms"), d.get("pipe", {}).get("frac"), d.get("bit_identical"))
PY
timeout 900 python scripts/fuzz_conv.py 300 67 > $O/fuzz_conv_$T.log 2>&1; echo "fuzz conv rc=$?"; tail -3 $O/fuzz_conv_$T.log | cut -c1-300
timeout 900 python scripts/fuzz_gemm.py 400 68 > $O/fuzz_gemm_$T.log 2>&1; echo "fuzz gemm rc=$?"; tail -3 $O/fuzz_gemm_$T.log | cut -c1-300
timeout 900 python scripts/bench_configs.py > $O/configs_$T.jsonl 2> $O/configs_$T.err; grep -i "C4\|conv\|C2\|C3" $O/configs_$T.jsonl | cut -c1-330
