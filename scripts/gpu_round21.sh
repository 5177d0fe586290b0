#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu21.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu21.log | cut -c1-200
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np, laser_amd
A = np.random.rand(300, 700).astype(np.float32); B = np.random.rand(700, 200).astype(np.float32)
C = np.zeros((300, 200), np.float32)
laser_amd.gemm_strided(300, 200, 700, 1.0, A, 700, 1, B, 200, 1, 0.0, C, 200, 1)
tA, tB = laser_amd.toTensor(A), laser_amd.toTensor(B)
tC = laser_amd.matmul(tA, tB, bias=laser_amd.toTensor(np.zeros((1, 200), np.float32)), activation="relu")
print("README example:", np.array_equal(tC.to_numpy(), np.maximum(C, 0)), tC.to_numpy()[:1, :3])
PY
