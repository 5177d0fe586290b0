#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_gpu18.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu18.log | cut -c1-200
OUT=$PWD/gpurun_out/prof_lds; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc.log 2>&1
python scripts/summarize_prof.py $OUT 2>/dev/null | grep -E "gemm_mfma|SQ_LDS|SQ_INSTS_LDS|SQ_ACTIVE" | cut -c1-160
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
timeout 900 python scripts/sweep_f32.py 8192 4 0,2,4 > gpurun_out/sweep18.log 2>&1; grep '"layout"' gpurun_out/sweep18.log | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
rows.sort(key=lambda r:(r['layout'],-r['tflops_med']))
for r in rows: print(r['n'], r['layout'], r['cfg'], r['mode'], r['ms_med'], r['tflops_med'])
"
