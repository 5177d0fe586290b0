"""world_size-2 gloo test of the row-panel sharded GEMM (laser_amd/distributed.py) on CPU.

The sharding / block-cyclic deal / pipelined all-gather logic is exercised with the local
multiply delegated to the CPU oracle (tests may use the oracle as the checker's arithmetic); on the
GPU box the same class runs with the HIP kernel and backend "nccl" (= RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, M, N, K, ppr, ret, gather="collective"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from laser_amd.distributed import ShardedGemm
        from oracle import oracle

        def local_gemm(A, B, out):
            out.copy_(torch.from_numpy(oracle.matmul(A.numpy(), B.numpy())))

        rng = np.random.default_rng(99)
        A = torch.from_numpy(rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32))
        B = torch.from_numpy(rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32))
        sg = ShardedGemm(M, N, K, torch.float32, None, None, ppr, local_gemm, gather=gather)
        C = sg.alloc_C()
        out = sg.run(sg.shard_A(A), B, C)
        want = oracle.matmul(A.numpy(), B.numpy())
        ok = np.array_equal(out.numpy(), want)
        rows = sg.local_rows()
        ret[rank] = (ok, len(rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,ppr", [((64, 48, 40), 1), ((200, 33, 50), 4), ((7, 5, 3), 4), ((1024, 64, 32), 2)])
def test_sharded_gemm_world2_gloo(shape, ppr):
    M, N, K = shape
    world = 2
    port = 29500 + (os.getpid() + M) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, M, N, K, ppr, ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world)), dict(ret)
    assert sum(ret[r][1] for r in range(world)) == M   # every row owned exactly once


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gemm_p2p_gather_gloo(world):
    """The point-to-point form of the gather of C (grouped isend / irecv pairs): same result on every rank."""
    M, N, K = 700, 48, 40
    port = 31500 + (os.getpid() + world) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, M, N, K, 3, ret, "p2p"), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world)), dict(ret)
    assert sum(ret[r][1] for r in range(world)) == M


def test_panel_plan_covers_rows_once():
    from laser_amd.distributed import make_plan
    for M in (1, 7, 255, 256, 8192, 65536, 65537, 1000):
        for world in (1, 2, 4, 8):
            for ppr in (1, 2, 4, 8):
                p = make_plan(M, world, ppr)
                seen = np.zeros(p.padded_M, dtype=int)
                for s in range(p.panels_per_rank):
                    lo, hi = p.slab(s)
                    for r in range(world):
                        start, valid = p.panel(s, r)
                        assert lo <= start and start + p.rows <= hi
                        seen[start:start + valid] += 1
                assert (seen[:M] == 1).all() and (seen[M:] == 0).all(), (M, world, ppr)
    p = make_plan(65536, 8, 4)
    assert p.rows == 2048 and p.padded_M == 65536


def _pin_worker(rank, world, port, ret):
    """ShardedGemm's default local product with the library mocked (no GPU here): the assembly tile pin (option "thread_asm_tile" = 2: per thread) must be
    in force around every local product of a multi-rank run, the caller's own value must come back afterwards -- also when a product
    raises -- and a forced compiler configuration must be bracketed the same way."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import laser_amd.primitives as prim
        from laser_amd import distributed as D
        state = {"thread_asm_tile": 7, "cfg": -1}        # 7: a value of the caller's own that must survive
        seen = []
        prim.get_option = lambda name: state[name]
        prim.set_option = lambda name, v: state.__setitem__(name, int(v))
        prim.get_f32_config = lambda: state["cfg"]
        prim.set_f32_config = lambda v: state.__setitem__("cfg", int(v))
        fail_after = {"n": None}

        def fake_matmul(A, B, alpha, beta, out):
            seen.append((state["thread_asm_tile"], state["cfg"]))
            if fail_after["n"] is not None and len(seen) > fail_after["n"]:
                raise RuntimeError("injected")
            out.copy_(A @ B)
        prim.matmul = fake_matmul
        M, N, K = 64, 16, 8
        g = torch.Generator().manual_seed(5)
        A, B = torch.rand((M, K), generator=g), torch.rand((K, N), generator=g)
        sg = D.ShardedGemm(M, N, K, torch.float32, None, None, 2)
        C = sg.alloc_C()
        out = sg.run(sg.shard_A(A), B, C)
        ok = torch.allclose(out, A @ B) and all(s_ == (D.SHARDED_ASM_TILE, -1) for s_ in seen) and len(seen) == 2
        ok = ok and state == {"thread_asm_tile": 7, "cfg": -1}
        # a forced compiler configuration (tuning sweeps): bracketed, the tile option untouched
        seen.clear()
        sg2 = D.ShardedGemm(M, N, K, torch.float32, None, None, 2, tile_config=3)
        sg2.run(sg2.shard_A(A), B, sg2.alloc_C())
        ok = ok and all(s_ == (7, 3) for s_ in seen) and state == {"thread_asm_tile": 7, "cfg": -1}
        # the library heuristic asked for explicitly: nothing is set
        seen.clear()
        sg3 = D.ShardedGemm(M, N, K, torch.float32, None, None, 2, tile_config=-1)
        sg3.run(sg3.shard_A(A), B, sg3.alloc_C())
        ok = ok and all(s_ == (7, -1) for s_ in seen) and state == {"thread_asm_tile": 7, "cfg": -1}
        # a product that raises (on every rank, before any collective of that step): the pin is undone
        seen.clear()
        fail_after["n"] = 0
        try:
            sg.run(sg.shard_A(A), B, sg.alloc_C())
            ok = False
        except RuntimeError:
            ok = ok and state == {"thread_asm_tile": 7, "cfg": -1}
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_gemm_tile_pin_is_bracketed_world2_gloo():
    world = 2
    port = 33500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pin_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
