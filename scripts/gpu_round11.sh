#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "torchrun world1 rc=$?"; tail -c 1500 gpurun_out/bench_torchrun1.json; tail -5 gpurun_out/bench_torchrun1.err
# exercise the world>1 code path (sharded run + RCCL all-gather + roofline branch) with a forced world flag on ONE gpu
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from laser_amd.distributed import ShardedGemm, _all_gather_rows
import laser_amd
sg = ShardedGemm(4096, 2048, 1024, torch.float32, torch.device("cuda:0"), None, 4)
A = (torch.rand((4096,1024), device="cuda")-0.5); B = (torch.rand((1024,2048), device="cuda")-0.5)
C = sg.alloc_C()
out = sg.run(sg.shard_A(A), B, C)
torch.cuda.synchronize()
print("plan", sg.plan, "err", (out.double() - A.double() @ B.double()).abs().max().item())
# the NCCL all-gather itself, in place, world 1
slab = torch.zeros((1024, 2048), device="cuda"); mine = slab[:1024]; mine.fill_(3.0)
w = _all_gather_rows(slab, mine, None); w.wait(); torch.cuda.synchronize()
print("all_gather_into_tensor in place ok:", bool((slab == 3).all()))
dist.destroy_process_group()
PY
