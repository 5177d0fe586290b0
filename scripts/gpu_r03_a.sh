#!/bin/bash
# round 3, GPU call A: parity of the assembly kernels through the product library, schedule sweep, product A/B, bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "f32_asm or full_size_8192 or race_screen" > gpurun_out/r03a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03a_pytest.log
tail -5 gpurun_out/r03a_pytest.log
timeout 600 python scripts/asm_probe.py scripts/asm_variants_v1.json --out gpurun_out/asm_probe_v1.jsonl 2>&1 | tail -30
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r03a_ab.log
import torch, laser_amd, json
n = 8192
A = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; B = (torch.rand((n, n), device="cuda") - 0.5) * 0.2; C = torch.zeros((n, n), device="cuda")
for size in (8192, 4096):
    a, b, c = A[:size, :size].contiguous(), B[:size, :size].contiguous(), C[:size, :size].contiguous()
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        res = {0: [], 1: []}
        for r in range(6):
            for asm in (0, 1):
                laser_amd.set_f32_asm(asm)
                laser_amd.matmul(a, b, 1, 0, c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4): laser_amd.matmul(a, b, 1, 0, c)
                e1.record(); torch.cuda.synchronize()
                if r: res[asm].append(e0.elapsed_time(e1) / 4)
        for asm in (0, 1):
            v = sorted(res[asm]); med = v[len(v) // 2]
            print(json.dumps({"n": size, "mode": "fast" if mode else "laser_order", "asm": asm, "ms": round(med, 4), "tflops": round(2 * size ** 3 / med / 1e9, 1), "frac": round(2 * size ** 3 / med / 1e9 / 157.3, 4)}), flush=True)
laser_amd.set_float_mode(0); laser_amd.set_f32_asm(1)
PY
timeout 600 python bench.py --no-single-process > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; tail -c 1500 gpurun_out/r03a_bench.json
