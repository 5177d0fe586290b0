#!/bin/bash
# round 5, call A: the whole GPU suite on the new sources (asm tile pin, C5-shape test, generic im2col / transposes, self-contained
# pre-pack, library-owned scratch pool), smoke, the bench line, the small-shape vendor yardstick, the LDS-conflict ablation, im2col alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-a}
O=gpurun_out/r05; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | tail -20
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 120 python scripts/im2col_probe.py > $O/im2col_probe_$T.jsonl 2> $O/im2col_probe_$T.err; cat $O/im2col_probe_$T.jsonl | cut -c1-260; tail -3 $O/im2col_probe_$T.err
timeout 500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; tail -1 $O/bench_$T.json | cut -c1-1500; tail -2 $O/bench_$T.err
timeout 200 python scripts/vendor_blas_probe.py small > $O/vendor_blas_small_$T.jsonl 2> $O/vendor_blas_small_$T.err; cat $O/vendor_blas_small_$T.jsonl; tail -2 $O/vendor_blas_small_$T.err
# LDS bank conflicts by instruction class: the headline kernels with their LDS stores / reads / A-stores / B-stores ablated
P=$PWD/$O/lds_conflict_pmc; rm -rf $P; mkdir -p $P
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $P/pmc -- python scripts/asm_probe.py scripts/asm_variants_lds.json --n 4096 --out $P/timing.jsonl > $P/pmc.log 2>&1
python scripts/summarize_prof.py $P lh_probe > $P/summary.md 2>&1; find $P -name "*_counter_collection.csv" -delete; find $P -name "*_agent_info.csv" -delete
grep -A8 "PMC" $P/summary.md | grep -E "PMC|SQ_" | head -60
