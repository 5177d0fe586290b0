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


def test_sharded_host_column_major_c_is_cut_by_columns(la):
    """with a column-major C the rows interleave in memory, so row ranges per GPU would stage
    overlapping spans and overwrite each other's finished rows.  The host form now cuts such a C into COLUMN ranges
    (B and C by columns, A replicated); a C whose rows AND columns interleave runs on one device.  Two and three device
    slots on the one GPU, every layout of C, against the oracle bit for bit."""
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(11)
    for (M, N, K) in [(1200, 900, 300), (700, 2100, 64), (33, 17, 9)]:
        A = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
        B = rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32)
        want = oracle.matmul(A, B)
        for devices in ([0, 0], [0, 0, 0]):
            Cf = np.full((N, M + 3), 5, dtype=np.float32)          # column-major C with a padded leading dimension
            Cv = Cf[:, :M].T
            assert Cv.strides == (4, 4 * (M + 3))
            la.gemm_strided_sharded(devices, M, N, K, 1.0, A, K, 1, B, N, 1, 0.0, Cv, 1, M + 3)
            assert np.array_equal(Cv, want), (M, N, K, devices, "column-major C")
            assert (Cf[:, M:] == 5).all(), "padding of the column-major C was touched"
            # beta != 0 reads C: same layout
            C0 = rng.uniform(-1, 1, (N, M)).astype(np.float32)
            Cv = C0.copy().T
            la.gemm_strided_sharded(devices, M, N, K, 0.5, A, K, 1, B, N, 1, 0.25, Cv, 1, M)
            one = C0.copy().T
            la.gemm_strided_sharded([0], M, N, K, 0.5, A, K, 1, B, N, 1, 0.25, one, 1, M)
            assert np.array_equal(Cv, one), (M, N, K, devices, "column-major C, beta != 0")
    # the routing knob takes the same path for an unchanged gemm_strided call
    M, N, K = 4096, 4096, 4096
    A = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
    B = rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32)
    want = la.matmul(A, B)
    Cv = np.zeros((N, M), dtype=np.float32).T
    try:
        la.set_shard_devices(0)
        la.gemm_strided(M, N, K, 1.0, A, K, 1, B, N, 1, 0.0, Cv, 1, M)
    finally:
        la.set_shard_devices(1)
    assert np.array_equal(Cv, want)


def test_sharded_dev_rccl_transport_at_one_rank(la):
    """GATHER_RCCL with ONE rank: dlopen of librccl.so, ncclCommInitAll, the in-place ncclAllGather pointer arithmetic and
    the bounded wait all execute on a single-GPU box (RCCL refuses two ranks on one device, so this is the widest it can
    run here).  Bit-identical to the single-GPU product; float32 and int32 (the ncclDataType_t constants)."""
    import torch
    for dtype in (np.float32, np.int32):
        M, N, K, ppd = 2048, 384, 640, 4
        A, B = _operands(M, N, K, dtype, seed=3)
        want = la.matmul(A, B)
        rows, ppd_used, padded = la.shard_plan(M, 1, ppd)
        C = torch.zeros((padded, N), dtype=want.dtype, device="cuda")
        la.gemm_strided_sharded_dev([0], M, N, K, 1, [la.shard_rows(A, 1, 0, ppd)], K, 1, [B], N, 1, 0, [C], N, ppd, la.GATHER_RCCL, 0)
        assert torch.equal(C[:M], want), dtype
    # a dense C is required by the flat slab sends
    with pytest.raises(la.LaserHipError):
        Cw = torch.zeros((padded, N + 8), device="cuda")
        A, B = _operands(M, N, K, np.float32, seed=3)
        la.gemm_strided_sharded_dev([0], M, N, K, 1.0, [la.shard_rows(A, 1, 0, ppd)], K, 1, [B], N, 1, 0.0, [Cw], N + 8, ppd, la.GATHER_RCCL, 0)


def test_sharded_tile_pin_is_per_call_not_global(la):
    """LASER_HIP_SHARD_PIN_TILE used to write the process-global f32 configuration and reset it to
    -1.  It is a per-thread override now: the caller's own settings survive the call, the result is unchanged.
    VERDICT r4 next #1(a): the pin selects the hand-scheduled 128x128x16 ASSEMBLY kernel (last_f32_asm() == 3 laser-order, 4 one
    chain), not the compiler-scheduled 128x128 configuration it used to force."""
    import torch
    M, N, K, ppd = 4096, 1024, 1100, 2
    A, B = _operands(M, N, K, np.float32, seed=9)
    want = la.matmul(A, B)
    rows, ppd_used, padded = la.shard_plan(M, 2, ppd)
    assert la.get_option("asm_tile") == -1
    for mode, asm_id in ((0, 3), (1, 4)):
        la.set_float_mode(mode)
        try:
            want_m = la.matmul(A, B)
            Cs = [torch.zeros((padded, N), device="cuda") for _ in range(2)]
            la.gemm_strided_sharded_dev([0, 0], M, N, K, 1.0, [la.shard_rows(A, 2, g, ppd) for g in range(2)], K, 1, [B, B], N, 1, 0.0, Cs, N,
                                        ppd, la.GATHER_PEER, la.SHARD_PIN_TILE)
            assert la.last_f32_asm() == asm_id, f"the pinned 128x128x16 assembly tile did not run (last_f32_asm {la.last_f32_asm()})"
            assert la.get_option("asm_tile") == -1, "the per-call pin leaked into the process-wide option"
            if mode == 0:
                for g in range(2):
                    assert torch.equal(Cs[g][:M], want), "laser-order result changed under the pinned tile"
            else:
                for g in range(2):
                    torch.testing.assert_close(Cs[g][:M], want_m, rtol=1e-4, atol=1e-5)
        finally:
            la.set_float_mode(0)
    # the caller's own forced compiler configuration still wins for ITS launches and is not clobbered by a pinned call
    try:
        la.set_f32_config(3)
        Cs = [torch.zeros((padded, N), device="cuda") for _ in range(2)]
        la.gemm_strided_sharded_dev([0, 0], M, N, K, 1.0, [la.shard_rows(A, 2, g, ppd) for g in range(2)], K, 1, [B, B], N, 1, 0.0, Cs, N,
                                    ppd, la.GATHER_PEER, la.SHARD_PIN_TILE)
        la.matmul(A, B)
        assert la.last_f32_config() == 3, "the caller's forced configuration was clobbered by the pin"
    finally:
        la.set_f32_config(-1)
    for g in range(2):
        assert torch.equal(Cs[g][:M], want)


def test_asm_tile_option_selects_the_tile_class(la):
    """Option "asm_tile" (what the per-GPU processes of laser_amd/distributed.py set around their local products): every class
    runs its own assembly kernel, laser-order results are the same bits under each."""
    import torch
    A, B = _operands(2048, 2048, 1100, np.float32, seed=21)
    want = la.matmul(A, B)
    try:
        for cls, asm_id in ((0, 1), (2, 3), (3, 31), (4, 13), (5, 47), (6, 51)):
            la.set_option("asm_tile", cls)
            got = la.matmul(A, B)
            assert la.last_f32_asm() == asm_id, (cls, la.last_f32_asm())
            assert torch.equal(got, want), cls
        la.set_option("asm_tile", -1)
        # "thread_asm_tile": the same pin for the calling thread only (what laser_amd/distributed.py sets around its local products,
        # ADVICE r5): this thread's launches take the class, another thread's keep the model's choice
        import threading
        la.matmul(A, B)
        free_choice = la.last_f32_asm()
        la.set_option("thread_asm_tile", 4)
        assert la.get_option("thread_asm_tile") == 4 and la.get_option("asm_tile") == -1
        got = la.matmul(A, B)
        assert la.last_f32_asm() == 13 and torch.equal(got, want)
        seen = {}

        def other():
            torch.cuda.set_device(0)
            seen["pin"] = la.get_option("thread_asm_tile")
            seen["equal"] = bool(torch.equal(la.matmul(A, B), want))
            seen["kernel"] = la.last_f32_asm()
        th = threading.Thread(target=other)
        th.start(); th.join()
        assert seen == {"pin": -2, "equal": True, "kernel": free_choice}, (seen, free_choice)
    finally:
        la.set_option("asm_tile", -1)
        la.set_option("thread_asm_tile", -2)


def test_c5_shape_eight_slots_every_panel_sampled_vs_oracle(la):
    """BASELINE configs[4] at its OWN shape (VERDICT r4 missing #2): fp32 65536 x 8192 x 8192 row-panel sharded over 8 device slots
    (all on this box's one GPU: every code path of the 8-GPU form -- 32 block-cyclic panels, 8 worker threads, 56 peer pushes per
    slab -- runs for real, only the wire is local; ~20 GiB of HBM), GATHER_PEER.  EVERY slot's gathered C is compared with slot 0's
    in full, and a slot that RECEIVED the rows is compared with the oracle bit for bit on 4096 sampled rows spanning all 32
    panels (128 rows per panel, at random offsets)."""
    import torch
    from oracle import oracle
    oracle.build()
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2**30:
        pytest.skip("needs ~20 GiB of free HBM")
    ndev, ppd, n = 8, 4, 8192
    M, N, K = n * ndev, n, n
    rows, ppd_used, padded = la.shard_plan(M, ndev, ppd)
    assert (rows, ppd_used, padded) == (2048, 4, M)
    gen = torch.Generator(device="cuda").manual_seed(505)
    B = torch.rand((K, N), generator=gen, device="cuda") * 0.2 - 0.1
    # A is made slot by slot (2 GiB in all): slot g's stack of its 4 panels
    Ap = [torch.rand((ppd_used * rows, K), generator=gen, device="cuda") * 0.2 - 0.1 for _ in range(ndev)]
    Cs = [torch.full((padded, N), float("nan"), device="cuda") for _ in range(ndev)]
    la.gemm_strided_sharded_dev([0] * ndev, M, N, K, 1.0, Ap, K, 1, [B] * ndev, N, 1, 0.0, Cs, N, ppd, la.GATHER_PEER, 0)
    torch.cuda.synchronize()
    assert la.last_f32_asm() != 0, "the local products did not run on the assembly kernels"
    for g in range(1, ndev):
        assert torch.equal(Cs[g], Cs[0]), f"slot {g}'s gathered C differs from slot 0's"
    rng = np.random.default_rng(5)
    Bh = B.cpu().numpy()
    for s in range(ppd_used):
        for g in range(ndev):
            start = (s * ndev + g) * rows
            loc = torch.from_numpy(np.sort(rng.choice(rows, 128, replace=False))).cuda()
            want = oracle.matmul(Ap[g][s * rows + loc].cpu().numpy(), Bh)
            got = Cs[(g + 3) % ndev][start + loc].cpu().numpy()   # a slot that RECEIVED these rows
            assert np.array_equal(got, want), f"panel (s={s}, slot={g}) differs from the oracle"


def test_bench_gpus2_without_torchrun_runs_through_the_c_abi():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the shape of the driver's command) must itself
    produce the line -- through laser_hip_gemm_strided_f32_sharded_dev in this process.  Two device slots on the one GPU
    here (LASER_BENCH_ONE_GPU=1); timings are meaningless, the contract and the self-check are what is tested."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["LASER_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "2048"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "GFLOP/s" and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["M"] == 4096 and "sharded_dev" in d["config"]["entry_point"]
    assert "roofline" in d and d["roofline"]["bound"] == "mfma"
