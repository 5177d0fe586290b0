#!/bin/bash
# LDS-DMA float32 kernel: parity (new test + the float32 suite) and timing against the register-staged kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dma or full_size_8192 or bit_exact_vs_oracle or fuzz or strided_and_transposed or split_tail" > $O/pytest_gpu_dma.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_dma.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_dma.log | tail -25
timeout 300 python - > $O/dma_probe.jsonl 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, laser_amd
from scripts.bench_configs import ev_time
for n in (8192, 4096):
    A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; C = torch.zeros((n, n), device="cuda")
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        r = {}
        for dma in (1, 0, 1, 0):
            laser_amd.set_f32_dma(dma)
            med, mn = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=7)
            r.setdefault(dma, []).append(med)
        laser_amd.set_f32_dma(1)
        print(json.dumps({"n": n, "mode": "laser_order" if mode == 0 else "fast", "dma_ms": [round(v, 4) for v in r[1]], "staged_ms": [round(v, 4) for v in r[0]],
                          "dma_tflops": round(2.0 * n ** 3 / min(r[1]) / 1e9, 1), "staged_tflops": round(2.0 * n ** 3 / min(r[0]) / 1e9, 1)}), flush=True)
laser_amd.set_float_mode(0)
PY
cat $O/dma_probe.jsonl
