#!/bin/bash
# fabric traffic and L2 hit rate of the 64x64 kernel (plain plan) against the 256x128 kernel on the same shape; each counter set its own
# rocprofv3 run under its own timeout
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export TMPDIR=/tmp
O=$R/gpurun_out/r04/small_tile_pmc; rm -rf $O; mkdir -p $O
N=${1:-3072}
for k in 12 0; do
  CMD="python scripts/shape_run.py $N $N $N 0 $k 1 30"
  timeout -k 5 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/k$k/fetch -- $CMD > $O/k$k.fetch.log 2>&1
  timeout -k 5 60 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/k$k/tcc -- $CMD > $O/k$k.tcc.log 2>&1
  python scripts/summarize_prof.py $O/k$k lh_ > $O/summary_k$k.md 2>&1
  grep "^| \|PMC\|dispatches" $O/summary_k$k.md | cut -c1-200
done
find $O -name "*_counter_collection.csv" -delete; find $O -name "*_agent_info.csv" -delete
