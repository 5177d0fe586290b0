#!/bin/bash
# round 2, GPU call 2: parity suite incl. the sharded entry points, bench line (new cpu_baseline protocol + single-process
# side measurement), C4 conv after the patch-stride fix, remaining heuristic-check shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_v2.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v2.log; tail -30 $O/pytest_gpu_v2.log
timeout 400 python bench.py > $O/bench_v2.json 2> $O/bench_v2.err; tail -1 $O/bench_v2.json; tail -3 $O/bench_v2.err
timeout 300 python scripts/conv_c4_run.py 10 > $O/conv_c4_v2.log 2>&1; cat $O/conv_c4_v2.log
timeout 400 python scripts/heuristic_check.py 1000x3000x2000,256x100352x1152,8192x512x8192,512x8192x4096,3000x3000x600,16384x1024x1024,3072x3072x3072,5000x5000x5000 > $O/heuristic_check_v2.jsonl 2> $O/heuristic_check_v2.err; python - <<'PY'
import json
for l in open('gpurun_out/r02/heuristic_check_v2.jsonl'):
    d=json.loads(l); print(d['shape'], d['mode'][:5], d['chosen'], 'cut',d['cut'], 'auto',d['auto_ms'],'nosplit',d['auto_nosplit_ms'],'best',d['best'],d['best_ms'],'TF',d['auto_tflops'],'ratio',d['auto_over_best'])
PY
