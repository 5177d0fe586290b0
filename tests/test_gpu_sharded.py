"""GPU tests of the multi-GPU paths with the REAL HIP local multiply:
  * the single-process sharded entry points of the C-ABI (laser_hip_gemm_strided_*_sharded[_dev]) with one device
    and with TWO device slots on the one physical GPU of the test box -- every code path of the N > 1 form (block-cyclic
    panels, per-rank threads and streams, events, peer copies of finished rows) runs for real, only the wire is local;
  * laser_amd.distributed.ShardedGemm (one process per GPU over torch.distributed) at world 1 over RCCL.
The gathered C must equal the single-GPU result BIT FOR BIT on every device slot: rows are independent units
(gemm.nim:160-176), there is no K split and hence no reduction."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import laser_amd
    laser_amd.lib()
    laser_amd.set_float_mode(0)
    laser_amd.set_f32_config(-1)
    return laser_amd


def _operands(M, N, K, dtype, seed=0):
    import torch
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        A = rng.uniform(-0.1, 0.1, (M, K)).astype(dtype)
        B = rng.uniform(-0.1, 0.1, (K, N)).astype(dtype)
    else:
        info = np.iinfo(dtype)
        A = rng.integers(info.min, info.max, (M, K), dtype=dtype)
        B = rng.integers(info.min, info.max, (K, N), dtype=dtype)
    return torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("shape,ppd", [((2048, 512, 1100), 4), ((1000, 384, 520), 3), ((4096, 1024, 1024), 2), ((70, 50, 30), 4)])
def test_sharded_dev_bit_identical_to_single_gpu(la, devices, shape, ppd):
    import torch
    M, N, K = shape
    ndev = len(devices)
    A, B = _operands(M, N, K, np.float32, seed=M + ndev)
    want = la.matmul(A, B)
    rows, ppd_used, padded = la.shard_plan(M, ndev, ppd)
    assert padded >= M and rows * ndev * ppd_used == padded
    for gather in (la.GATHER_PEER, la.GATHER_NONE):
        Ap = [la.shard_rows(A, ndev, g, ppd) for g in range(ndev)]
        Bs = [B.clone() for _ in range(ndev)]
        Cs = [torch.full((padded, N), float("nan"), device="cuda") for _ in range(ndev)]
        la.gemm_strided_sharded_dev(devices, M, N, K, 1.0, Ap, K, 1, Bs, N, 1, 0.0, Cs, N, ppd, gather, 0)
        for g in range(ndev):
            if gather == la.GATHER_PEER or ndev == 1:
                assert torch.equal(Cs[g][:M], want), (devices, shape, g, "gathered C differs from the single-GPU result")
            else:   # no gather: only slot g's own panels are written
                for s in range(ppd_used):
                    start = (s * ndev + g) * rows
                    valid = max(0, min(rows, M - start))
                    assert torch.equal(Cs[g][start:start + valid], want[start:start + valid])
    # laser-order result vs the oracle on a sample of rows (ties the sharded path to the reference arithmetic)
    from oracle import oracle
    oracle.build()
    r = slice(0, min(M, 128))
    assert np.array_equal(want[r].cpu().numpy(), oracle.matmul(A[r].cpu().numpy(), B.cpu().numpy()))


def test_sharded_dev_padded_rows_beta_and_other_dtypes(la):
    """rowStrideC > N (the 2-D peer copy), alpha / beta != (1, 0) (each slot reads its own copy of its rows), f64 / i32."""
    import torch
    M, N, K, ndev, ppd = 1536, 200, 700, 2, 2
    for dtype, alpha, beta in ((np.float32, 0.5, 0.25), (np.float64, 1.0, 0.0), (np.int32, 3, -2)):
        A, B = _operands(M, N, K, dtype, seed=7)
        C0 = torch.from_numpy(np.random.default_rng(1).integers(-50, 50, (M, N)).astype(dtype)).cuda()
        want = la.matmul(A, B, alpha, beta, C0.clone())
        rows, ppd_used, padded = la.shard_plan(M, ndev, ppd)
        ld = N + 24
        Cs = []
        for g in range(ndev):
            buf = torch.zeros((padded, ld), dtype=C0.dtype, device="cuda")
            buf[:M, :N] = C0
            buf[:, N:] = 7
            Cs.append(buf)
        Ap = [la.shard_rows(A, ndev, g, ppd) for g in range(ndev)]
        la.gemm_strided_sharded_dev([0, 0], M, N, K, alpha, Ap, K, 1, [B, B], N, 1, beta, Cs, ld, ppd, la.GATHER_PEER, 0)
        for g in range(ndev):
            assert torch.equal(Cs[g][:M, :N], want), (dtype, g)
            assert (Cs[g][:, N:] == 7).all(), "the gather touched the padding columns of C"


def test_sharded_host_pointers_bit_identical(la, ):
    """The drop-in form: gemm_strided's parameter list after the device list, host pointers, one row range per GPU slot
    (two slots on the one GPU here: two host threads through the per-device host pipeline)."""
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(3)
    for (M, N, K) in [(3000, 700, 900), (513, 64, 40), (5, 9, 3)]:
        A = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
        B = rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32)
        want = oracle.matmul(A, B)
        for devices in ([0], [0, 0]):
            got = la.matmul_sharded(A, B, devices)
            assert np.array_equal(got, want), (M, N, K, devices)
        Bt = np.ascontiguousarray(B.T).T            # transposed-B strides through the sharded entry point
        C = np.full((M, 2 * N), 9, dtype=np.float32)[:, ::2]
        la.gemm_strided_sharded([0, 0], M, N, K, 1.0, A, K, 1, Bt, 1, K, 0.0, C, 2 * N, 2)
        assert np.array_equal(C, want)
    Ai = rng.integers(-2**31, 2**31 - 1, (700, 90), dtype=np.int32)
    Bi = rng.integers(-2**31, 2**31 - 1, (90, 50), dtype=np.int32)
    assert np.array_equal(la.matmul_sharded(Ai, Bi, [0, 0]), oracle.matmul(Ai, Bi))


def test_sharded_errors_and_routing_knob(la):
    import torch
    n = torch.cuda.device_count()
    A = np.ones((8, 8), np.float32)
    with pytest.raises(la.LaserHipError):
        la.matmul_sharded(A, A, [n])                       # device ordinal out of range
    with pytest.raises(la.LaserHipError):
        la.gemm_strided_sharded_dev([0], 8, 8, 8, 1.0, [torch.ones(8, 8, device="cuda")], 8, 1, [torch.ones(8, 8, device="cuda")], 8, 1,
                                    0.0, [torch.ones(8, 8, device="cuda")], 4, 1, la.GATHER_PEER, 0)   # rowStrideC < N
    try:   # routing knob: small calls never shard; a large one is cut over n + 1 devices -> fails loudly on this box
        la.set_shard_devices(n + 1)
        assert la.get_shard_devices() == n + 1
        assert np.array_equal(la.matmul(A, A), A @ A)
        big = np.zeros((1024 * (n + 1), 8192), np.float32)   # M >= 1024 per device and M.N.K >= 2^36: gets routed
        with pytest.raises(la.LaserHipError):
            la.matmul(big, np.zeros((8192, 4096), np.float32))
        la.set_shard_devices(0)                            # "every visible GPU"
        rng = np.random.default_rng(0)
        a = rng.uniform(-0.1, 0.1, (1024 * max(n, 1), 4096)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, (4096, 4096)).astype(np.float32)
        got = la.matmul(a, b)
        la.set_shard_devices(1)
        assert np.array_equal(got, la.matmul(a, b))
    finally:
        la.set_shard_devices(1)


def test_sharded_gemm_class_world1_rccl_real_hip(la):
    """laser_amd.distributed.ShardedGemm with the REAL local multiply over RCCL at world 1 (the all-gather with one rank
    is the identity): gathered C == single-GPU result, bit for bit."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import laser_amd
from laser_amd.distributed import ShardedGemm
g = torch.Generator(device="cuda").manual_seed(5)
M, N, K = 2304, 640, 1300
A = (torch.rand((M, K), generator=g, device="cuda") - 0.5) * 0.2
B = (torch.rand((K, N), generator=g, device="cuda") - 0.5) * 0.2
sg = ShardedGemm(M, N, K, torch.float32, torch.device("cuda", 0), None, 4)
C = sg.alloc_C()
out = sg.run(sg.shard_A(A), B, C)
torch.cuda.synchronize()
assert torch.equal(out, laser_amd.matmul(A, B)), "ShardedGemm differs from the single-GPU product"
dist.destroy_process_group()
print("SHARDED_WORLD1_OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "SHARDED_WORLD1_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
