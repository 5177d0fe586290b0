cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r04/conv_direct_pmc2; rm -rf $O; mkdir -p $O
CMD="python scripts/probes/conv_direct_loop.py 1 64"
timeout -k 5 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 5 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
timeout -k 5 100 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/tcp -- $CMD > $O/tcp.log 2>&1; echo "tcp rc=$?"
timeout -k 5 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tcc -- $CMD > $O/tcc.log 2>&1; echo "tcc rc=$?"
python scripts/summarize_prof.py $O conv_direct > $O/summary.md 2>&1
find $O -name "*_counter_collection.csv" -delete; find $O -name "*_agent_info.csv" -delete
grep -v "^$" $O/summary.md | grep "^|\|PMC" | cut -c1-160
