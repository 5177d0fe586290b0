#!/bin/bash
# full validation of the tree on one MI355X (gpurun): whole GPU suite, smoke, headline profile (+ the traffic file made on the same
# tree), the bench line, rocprofv3 evidence for C3 / C4 (warm-only averages), the N > 1 line formats on one GPU, every BASELINE config line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-final}
O=gpurun_out/r06; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_$T.log; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed" | tail -12
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1000 bash scripts/gpu_profile_bench.sh default > /dev/null 2>&1; rm -rf $O/rocprof_bench_default; cp -r gpurun_out/prof_default $O/rocprof_bench_default; head -12 $O/rocprof_bench_default/summary.md; cat $O/rocprof_bench_default/pmc_traffic.json | cut -c1-300 | head -14
cp $O/rocprof_bench_default/pmc_traffic.json profiles/pmc_traffic.json
timeout 500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err; tail -1 $O/bench_$T.json | cut -c1-1500; tail -2 $O/bench_$T.err
SKIP=60 timeout 600 bash scripts/gpu_profile_cmd.sh c3 python scripts/c3_run.py 40 > /dev/null 2>&1; rm -rf $O/rocprof_c3; cp -r gpurun_out/prof_c3 $O/rocprof_c3; head -8 $O/rocprof_c3/summary.md
SKIP=60 timeout 600 bash scripts/gpu_profile_cmd.sh c4 python scripts/conv_c4_run.py 10 > /dev/null 2>&1; rm -rf $O/rocprof_c4; cp -r gpurun_out/prof_c4 $O/rocprof_c4; head -9 $O/rocprof_c4/summary.md
LASER_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --size 4096 > $O/bench_gpus2_one_process_$T.json 2> $O/bench_gpus2_one_process_$T.err; tail -1 $O/bench_gpus2_one_process_$T.json | cut -c1-700
LASER_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 3 --warmup 1 --size 4096 --panels-per-rank 4 --gather collective --no-single-process > $O/bench_gpus2_torchrun_$T.json 2> $O/bench_gpus2_torchrun_$T.err; grep '^{' $O/bench_gpus2_torchrun_$T.json | tail -1 | cut -c1-1500
timeout 900 python scripts/bench_configs.py > $O/configs_$T.jsonl 2> $O/configs_$T.err; wc -l $O/configs_$T.jsonl; tail -2 $O/configs_$T.err
timeout 100 python scripts/im2col_probe.py > $O/im2col_probe_$T.jsonl 2> /dev/null; cut -c1-200 $O/im2col_probe_$T.jsonl | head -3
