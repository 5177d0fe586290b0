#!/bin/bash
# round 4, call: new 128x128x32 kernels in the sweep + counters of the 64x64 and 128x128 kernels on the reference's published shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-h}
O=gpurun_out/r04; mkdir -p $O
timeout 300 python scripts/plan_sweep.py f32 mid > $O/plan_sweep_f32_mid_$T.jsonl 2> $O/plan_sweep_$T.err; echo "sweep rc=$?"
timeout 400 bash scripts/gpu_profile_cmd.sh r04_1920_64x64 python scripts/shape_run.py 1920 1920 1920 0 12 1 60 > /dev/null 2>&1; cp -r gpurun_out/prof_r04_1920_64x64 $O/rocprof_1920_64x64; head -40 $O/rocprof_1920_64x64/summary.md
timeout 400 bash scripts/gpu_profile_cmd.sh r04_1920_128x128 python scripts/shape_run.py 1920 1920 1920 0 2 1 60 > /dev/null 2>&1; cp -r gpurun_out/prof_r04_1920_128x128 $O/rocprof_1920_128x128; head -40 $O/rocprof_1920_128x128/summary.md
