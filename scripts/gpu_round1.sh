#!/bin/bash
# first GPU session: smoke, parity tests, tile sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(nproc; lscpu | grep -E "Model name|Flags" | cut -c1-300; rocm-smi --showproductname 2>/dev/null | head -8) > gpurun_out/host.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
timeout 600 python scripts/sweep_f32.py 8192,4096 3 > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"; tail -70 gpurun_out/sweep.log
