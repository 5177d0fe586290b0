#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "transpose or tensor or nchw" > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu10.log | cut -c1-200
timeout 900 python scripts/sweep_f32.py 512,1024,1536,2048,3072 4 > gpurun_out/sweep10.log 2>&1; echo "sweep rc=$?"; grep '"nn"' gpurun_out/sweep10.log | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
rows.sort(key=lambda r:(r['n'], r['mode'], -r['tflops_med']))
for r in rows: print(r['n'], r['mode'], r['cfg'], r['ms_med'], r['tflops_med'])
"
for n in 512 1024 1536 2048 3072; do python - <<PY
import torch, laser_amd, sys
n=$n
A=(torch.rand((n,n),device='cuda')-0.5)*0.2; B=(torch.rand((n,n),device='cuda')-0.5)*0.2; C=torch.zeros((n,n),device='cuda')
laser_amd.set_f32_config(-1)
for mode in (0,1):
    laser_amd.set_float_mode(mode)
    for _ in range(5): laser_amd.matmul(A,B,1,0,C)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): laser_amd.matmul(A,B,1,0,C)
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
    print('auto', n, 'laser' if mode==0 else 'fast', round(ms,4), round(2*n**3/ms/1e9,1))
PY
done 2>&1 | grep auto
