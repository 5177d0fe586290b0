#!/bin/bash
# kernel-trace + one PMC pass of the C4 convolution (scripts/conv_c4_run.py): durations of the main launch and of the pixel tail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_tail; rm -rf $O; mkdir -p $O gpurun_out/r05
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python scripts/conv_c4_run.py 10 > $O/stats.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc -- python scripts/conv_c4_run.py 10 > $O/pmc.log 2>&1
python scripts/summarize_prof.py $O --skip 60 > gpurun_out/r05/prof_tail_summary.md 2>&1
find $O -name "*.csv" -delete
head -14 gpurun_out/r05/prof_tail_summary.md; grep -A14 "conv3x3_tail" gpurun_out/r05/prof_tail_summary.md | tail -16; tail -3 $O/stats.log
