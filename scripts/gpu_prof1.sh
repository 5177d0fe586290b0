#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -i -E "mfma|GRBM_GUI|SQ_WAIT|SQ_BUSY|SQ_WAVE|LDS_BANK|LDS_IDX|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|SQ_INSTS|SQ_ACTIVE|SQ_INST_CYCLES" $OUT/counters_list.txt | cut -c1-160 | sort -u | head -120 > $OUT/counters_grep.txt
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode fast --cfg 7"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -30
for f in $(find $OUT -name "*.csv" | head -30); do echo "== $f"; head -4 $f | cut -c1-600; done
tail -3 $OUT/stats.log $OUT/pmc1.log $OUT/pmc2.log
