#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-k}
O=gpurun_out/r04; mkdir -p $O
F='hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:'
timeout 900 python -m pytest tests/test_gpu_scheduler.py -q --timeout 600 -k "prologue" > $O/prologue_tests_$T.log 2>&1; echo "prologue rc=$?"; grep -v "$F" $O/prologue_tests_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | tail -12
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_$T.log 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu_$T.log | grep -E "^FAILED|^ERROR|passed|failed|Error" | tail -30
python - <<'PY'
import sys, json, time; sys.path.insert(0, ".")
import torch, laser_amd, numpy as np
# cost of the fused prologue: 4096^3, plain vs relu(A) in-kernel vs relu(A) as a separate elementwise pass + plain
n = 4096
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
B = (torch.rand((n, n), generator=g, device="cuda") - 0.5) * 0.2
C = torch.zeros((n, n), device="cuda")
def t(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mode in (0, 1):
    laser_amd.set_float_mode(mode)
    rec = {"shape": "4096^3", "mode": "laser_order" if mode == 0 else "fast"}
    rec["plain_ms"] = round(t(lambda: laser_amd.matmul(A, B, 1, 0, C)), 4)
    rec["fused_relu_a_ms"] = round(t(lambda: laser_amd.matmul(A, B, 1, 0, C, pre=laser_amd.PRE_RELU_A)), 4); rec["kernel_a"] = laser_amd.last_f32_asm()
    rec["fused_relu_ab_ms"] = round(t(lambda: laser_amd.matmul(A, B, 1, 0, C, pre=laser_amd.PRE_RELU_A | laser_amd.PRE_RELU_B)), 4)
    R = torch.empty_like(A)
    rec["separate_pass_relu_a_ms"] = round(t(lambda: (torch.clamp(A, min=0, out=R), laser_amd.matmul(R, B, 1, 0, C))), 4)
    print(json.dumps(rec))
laser_amd.set_float_mode(0)
PY
