#!/bin/bash
# round 5, call C: what a 36-K-tile 256x128 tile kernel can reach at all (the GEMM twin of C4's main launch: 768 tiles = 3 exact rounds,
# K = 1152), the im2col band sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-c}
O=gpurun_out/r05; mkdir -p $O
for sh in "8192 3072 1152" "8192 3072 2304" "8192 3072 4608" "8192 3072 9216"; do
  timeout 100 python scripts/shape_run.py $sh 0 0 1 60 >> $O/gemm_twin_of_c4_$T.jsonl 2>> $O/gemm_twin_$T.err
  timeout 100 python scripts/shape_run.py $sh 1 8 1 60 >> $O/gemm_twin_of_c4_$T.jsonl 2>> $O/gemm_twin_$T.err
done
cat $O/gemm_twin_of_c4_$T.jsonl
timeout 100 python scripts/conv_c4_run.py 20 | tee $O/conv_c4_$T.log
timeout 200 python scripts/im2col_probe.py bands > $O/im2col_bands_$T.jsonl 2> $O/im2col_bands_$T.err; cut -c1-400 $O/im2col_bands_$T.jsonl; tail -2 $O/im2col_bands_$T.err
