#!/bin/bash
# round 2, GPU call 11: ragged-by-a-few peeling (parity + timing)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "ragged or skinny or split_tail or fuzz or strided_and_transposed" > $O/pytest_gpu_v11_ragged.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v11_ragged.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v11_ragged.log | tail -30
timeout 300 python scripts/ragged_probe.py > $O/ragged_probe_v1.jsonl 2>&1; cat $O/ragged_probe_v1.jsonl
