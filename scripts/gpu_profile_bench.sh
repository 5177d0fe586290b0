#!/bin/bash
# rocprofv3 evidence for bench.py's headline kernel: kernel-trace stats, then PMC passes (each in its
# own run; never combined with sys/hip/hsa tracing).  usage: gpu_profile_bench.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-default}; shift
OUT=$PWD/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-single-process --no-side-configs $*"
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
timeout -k 5 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout -k 5 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1
python scripts/summarize_prof.py $OUT --skip 10 > $OUT/summary.md 2>&1
python scripts/update_pmc_traffic.py $OUT "profiles/r06/rocprof_bench_$TAG/summary.md" > /dev/null 2>&1
for f in $(find $OUT/stats -name "*_kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
grep -h '"metric"' $OUT/*.log | head -3 > $OUT/bench_lines_under_profiler.jsonl
# keep the merged-back payload small: raw per-dispatch CSVs are large
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*_counter_collection.csv" -delete
cat $OUT/summary.md
