#!/bin/bash
# round 2, GPU call 8: 2-D host pipeline (parity + end-to-end timing A/B)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "host_pointer or pinned or sharded" > $O/pytest_gpu_v8_host.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_v8_host.log; grep -v "hip_runtime\|nodiscard\|hipError_t\|~~~\|^ *[0-9]* |\|^In file\|^ *from\|note:" $O/pytest_gpu_v8_host.log | tail -30
timeout 300 python scripts/host_pipeline_probe.py > $O/host_pipeline_v1.jsonl 2>&1; cat $O/host_pipeline_v1.jsonl
