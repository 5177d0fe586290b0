#!/bin/bash
# round 6, call F: conv geometry A/B (assembly loader vs compiler-scheduled loaders), conv fuzz after the oW == 1 fix, bench-contract tests, the N = 8 one-GPU-hook line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-f}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/conv_geometry_ab.py 32 > $O/conv_geometry_ab_$T.jsonl 2> $O/conv_geometry_ab_$T.err; cut -c1-700 $O/conv_geometry_ab_$T.jsonl; tail -3 $O/conv_geometry_ab_$T.err
timeout 900 python scripts/fuzz_conv.py 500 62 > $O/fuzz_conv_$T.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz_conv_$T.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q --timeout 900 > $O/pytest_bench_$T.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_bench_$T.log | cut -c1-300
LASER_BENCH_ONE_GPU=1 timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 > $O/bench_gpus8_one_process_one_gpu_hook_v2.json 2> $O/bench_gpus8_one_process_one_gpu_hook_v2.err; echo "n8 rc=$?"; cut -c1-1200 $O/bench_gpus8_one_process_one_gpu_hook_v2.json; tail -3 $O/bench_gpus8_one_process_one_gpu_hook_v2.err
