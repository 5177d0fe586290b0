"""CPU-side checks of host logic that needs no GPU: the LDS bank model of the k-quad image, the Tensor twin's
metadata arithmetic, and the loud failure of every device-touching call on a GPU-less host."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_k_quad_image_is_bank_conflict_free_in_the_model():
    """scripts/kq_bank_check.py replays the three LDS access patterns of the k-quad image against the bank rules of
    MI355X_MICROARCH.md; the swizzle + row swap the kernel uses must come out at zero extra cycles for BK = 16 and 32
    (the variant without the row swap must not: that is the conflict SQ_LDS_BANK_CONFLICT showed on hardware)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kq_bank_check.py")], capture_output=True, text=True, check=True)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    kernel = [l for l in lines if "kernel (kq_swz + kq_row)" in l]
    assert len(kernel) == 2 and all(l.rstrip().endswith("total 0") for l in kernel), r.stdout
    plain16 = [l for l in lines if l.startswith("16 no row swap")]
    assert plain16 and not plain16[0].rstrip().endswith("total 0")


def test_tensor_metadata_matches_numpy_without_touching_a_device():
    from laser_amd.tensor import Tensor, _row_major_strides
    for shape in [(3, 4, 5), (7,), (2, 1, 6, 1), (), (4, 0, 3)]:
        strides, size = _row_major_strides(shape)
        ref = np.zeros(shape, np.float32)
        assert size == ref.size and strides == tuple(s // 4 for s in ref.strides) or ref.size == 0
        t = Tensor(shape, strides, 0, None, np.float32)            # metadata only: storage is never dereferenced here
        assert t.rank == ref.ndim and t.size == ref.size and t.is_C_contiguous()
    t = Tensor((6, 8, 10), (80, 10, 1), 0, None, np.int64)
    ref = np.arange(480).reshape(6, 8, 10)
    assert t[-1, -2, -3].offset == 5 * 80 + 6 * 10 + 7 and t[-1, -2, -3].shape == ()
    for idx in [(slice(1, 5, 2), 3, slice(None, None, -1)), (2,), (slice(None), slice(2, 8, 3))]:
        v = t[idx]
        r = ref[idx]
        assert v.shape == r.shape and v.strides == tuple(s // ref.itemsize for s in r.strides)
        assert v.offset == (r.__array_interface__["data"][0] - ref.__array_interface__["data"][0]) // ref.itemsize
        assert v.is_C_contiguous() == (r.flags["C_CONTIGUOUS"] or r.size <= 1)
    assert t.transpose(2, 0, 1).strides == tuple(s // 8 for s in ref.transpose(2, 0, 1).strides)
    with pytest.raises(ValueError):
        Tensor((1,) * 7, (1,) * 7, 0, None, np.float32)            # LASER_MAXRANK
    with pytest.raises(IndexError):
        t[6]


def test_tensor_allocation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import laser_amd
    with pytest.raises(laser_amd.LaserHipError) as e:
        laser_amd.newTensor(np.float32, 4, 4)
    assert e.value.code == 3
    with pytest.raises(laser_amd.LaserHipError):
        laser_amd.toTensor([[1.0, 2.0], [3.0, 4.0]])


def test_shard_plan_matches_the_per_process_plan():
    """laser_hip_shard_plan (single-process sharded entry points) deals rows exactly like
    laser_amd.distributed.make_plan (one process per GPU), so both paths lay C out identically."""
    import itertools
    import laser_amd
    from laser_amd.distributed import make_plan
    for M, n, p in itertools.product([1, 5, 255, 256, 1000, 4100, 8192, 65536, 65537, 100000], [1, 2, 3, 4, 8], [1, 2, 4, 8, 16]):
        b = make_plan(M, n, p)
        assert laser_amd.shard_plan(M, n, p) == (b.rows, b.panels_per_rank, b.padded_M), (M, n, p)
    rows, ppd, padded = laser_amd.shard_plan(65536, 8, 4)
    assert (rows, ppd, padded) == (2048, 4, 65536)     # BASELINE configs[4]: 4 panels of 2048 rows per GPU


def _balanced_limbs(a, nlimbs):
    """The limb split the integer kernels use (gemm_i32_mfma.hip / gemm_i64_mfma.hip, limb_planes.h): bytes of
    (a + 0x..808080) ^ 0x..808080 read as int8 -- balanced base-256 digits, top digit any representative mod 256."""
    bits = 8 * nlimbs
    mask = (1 << bits) - 1
    bias = int("00" + "80" * (nlimbs - 1), 16)
    out = np.zeros((nlimbs,) + a.shape, dtype=np.int64)
    flat = [((int(v) & mask) + bias & mask) ^ bias for v in a.reshape(-1)]
    for p in range(nlimbs):
        d = np.array([(v >> (8 * p)) & 0xFF for v in flat], dtype=np.int64).reshape(a.shape)
        out[p] = np.where(d >= 128, d - 256, d)
    return out


@pytest.mark.parametrize("dtype,nlimbs", [(np.int32, 4), (np.int64, 8)])
def test_limb_decomposition_identity(dtype, nlimbs):
    """The arithmetic the int8-limb kernels rest on, replayed in numpy: a == sum_p s_p 256^p (mod 2^n) with s_p in
    [-128, 127]; a.b == sum_{p+q<n/8} s_p(a) s_q(b) 256^(p+q) (mod 2^n); and with K <= 8192 per accumulation chunk every
    per-power partial sum G_s stays inside int32 (|G_s| <= (s+1) * K * 2^14 <= 2^30), so nothing depends on overflow."""
    rng = np.random.default_rng(5)
    info = np.iinfo(dtype)
    bits = 8 * nlimbs
    M, N, K = 9, 7, 64
    A = rng.integers(info.min, info.max, (M, K), dtype=dtype)
    B = rng.integers(info.min, info.max, (K, N), dtype=dtype)
    A[0, :4] = [info.min, info.max, -1, 0]
    B[:4, 0] = [info.max, info.min, 127, -129]
    la, lb = _balanced_limbs(A, nlimbs), _balanced_limbs(B, nlimbs)
    assert la.min() >= -128 and la.max() <= 127
    # digits recompose the value mod 2^bits
    rec = sum(int(la[p][0, 1]) * 256 ** p for p in range(nlimbs)) % (1 << bits)
    assert rec == int(A[0, 1]) % (1 << bits)
    # per-power partial products, combined exactly as the kernels' epilogues do
    want = (A.astype(object) @ B.astype(object)) % (1 << bits)
    got = np.zeros((M, N), dtype=object)
    for s in range(nlimbs):
        G = sum(la[p] @ lb[s - p] for p in range(s + 1))           # int64 numpy: exact
        assert np.abs(G).max() <= (s + 1) * K * 2 ** 14
        got = (got + G.astype(object) * 256 ** s) % (1 << bits)
    assert (got == want).all()
    # the bound that sizes the launch chunk / fold interval
    assert 8 * 8192 * 2 ** 14 == 2 ** 30 and 4 * 8192 * 2 ** 14 == 2 ** 29      # int64: 8 pairs at most per power; int32: 4


# ds_read_b128 lane groups of MI355X_MICROARCH.md's LDS table (one LDS cycle per group when its 16 lanes hit 16 distinct
# 16-byte slots of a 256-byte bank line)
_R128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
_R128_GROUPS += [[l + 32 for l in g] for g in _R128_GROUPS]


def _b128_extra_cycles(addr_of_lane):
    """Extra LDS cycles of one wave-wide ds_read_b128 whose lane l reads 16 bytes at addr_of_lane(l)."""
    extra = 0
    for grp in _R128_GROUPS:
        slots = {}
        for l in grp:
            a = addr_of_lane(l)
            assert a % 16 == 0
            slots.setdefault((a // 16) % 16, set()).add(a)
        extra += max(len(v) for v in slots.values()) - 1
    return extra


def test_dma_fed_lds_layouts_are_bank_conflict_free_in_the_model():
    """The layouts filled by LDS-DMA, replayed against the same bank rules: the int64 limb kernel's 32-byte plane rows
    (chunk c of row r at slot c ^ ((r>>3)&1), gemm_i64_mfma.hip), a 128-byte-row image (chunk c of row x at slot
    c ^ ((x>>1)&7)) and the int32 limb planes.  Without their swizzles the patterns collide."""
    # int64 limb planes: lane (lo = l%32, hi = l//32) reads chunk hi of row lo
    swz = lambda l: (l % 32) * 32 + 16 * ((l // 32) ^ (((l % 32) >> 3) & 1))
    raw = lambda l: (l % 32) * 32 + 16 * (l // 32)
    assert _b128_extra_cycles(swz) == 0 and _b128_extra_cycles(raw) > 0
    # rows of 128 bytes: every lane of a group reads logical chunk c of its row
    for c in range(8):
        swz = lambda l, c=c: (l % 32) * 128 + 16 * (c ^ (((l % 32) >> 1) & 7))
        raw = lambda l, c=c: (l % 32) * 128 + 16 * c
        assert _b128_extra_cycles(swz) == 0, c
        assert _b128_extra_cycles(raw) > 0, c
    # int32 limb planes (gemm_i32_mfma.hip): 64-byte rows, chunk c of row r at slot c ^ ((r>>2)&3), lane reads chunk 2*ks + hi
    for ks in range(2):
        swz = lambda l, ks=ks: (l % 32) * 64 + 16 * ((2 * ks + l // 32) ^ (((l % 32) >> 2) & 3))
        assert _b128_extra_cycles(swz) == 0, ks


def test_bench_kernel_names_match_the_launcher_table():
    """bench.py names the kernel of its `roofline` object from laser_hip_get_option("last_f32_asm") = 1 + index into kKernels
    (gemm_f32_asm.cpp): the two tables must list the same symbols in the same order, and every symbol must be one the Makefile
    assembles into the embedded code object."""
    import re
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "laser_amd", "csrc", "gemm_f32_asm.cpp")).read()
    table = src[src.index("const KernelInfo kKernels[kNumKernels] = {"):src.index("// plain kernel -> its `_pre` variant")]
    symbols = re.findall(r'\{"(lh_[a-z0-9_x]+)"', table)
    assert len(symbols) == int(re.search(r"constexpr int kNumKernels = (\d+);", src).group(1))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.ASM_KERNEL_SYMBOLS == symbols
    mk = open(os.path.join(root, "laser_amd", "csrc", "Makefile")).read()
    for sym in symbols:
        assert re.search(r"\b" + sym + r"\b", mk), sym


def test_committed_traffic_figure_matches_the_shipped_kernels():
    """`roofline.traffic` of bench.py is a COMMITTED measurement (profiles/pmc_traffic.json), quoted only while the guard recorded with it
    -- the sha of the two headline kernels' generated assembly text + the launch plans they were measured under -- matches the tree.
    This test fails when the headline kernels changed without the counters being collected again (scripts/gpu_profile_bench.sh), which
    is when the bench line would silently go back to `traffic: null`."""
    import json
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert set(d["plans"]) == {"laser_order", "fast"}
    assert d["kernel_source_sha16"] == bench.kernel_source_sha16(d["plans"])
    for mode in ("laser_order", "fast"):
        assert d[mode]["bytes_per_launch"] == int(round((2 * d[mode]["fetch_size_kib"] + d[mode]["write_size_kib"]) * 1024))
        assert bench.pmc_traffic(mode, d["plans"][mode]) == d[mode]
        assert bench.pmc_traffic(mode, dict(d["plans"][mode], wgs=1)) is None      # another launch plan: the figure is not quoted


def test_launch_plans_scale_with_the_device_cu_count():
    """VERDICT r5 missing #4: 256 CUs / 8 XCDs were compile-time constants and a partitioned MI355X (CPX 32 / QPX 64 / DPX 128 CUs)
    stepped off every assembly kernel.  The launcher now plans with the device's own CU count; laser_hip_plan_f32 shows the choice
    without a device (the GPU twin of reading gemm_tiling.nim:276-341's partition for a shape)."""
    import laser_amd as la
    head = la.plan_f32(8192, 8192, 8192, True, 256)
    assert head == dict(kernel=1, plan="strided", wgs=256, slices=1, tiles=2048, tiles_m=32, tiles_n=64, slots=256)
    for cus in (32, 64, 128, 256):
        for shape in [(8192, 8192, 8192), (4096, 4096, 4096), (4100, 4100, 4100), (3072, 3072, 3072), (2048, 2048, 2048), (1920, 1920, 1920)]:
            for laser in (True, False):
                p = la.plan_f32(*shape, laser, cus)
                assert p["kernel"] != 0, (cus, shape, "a partitioned device must keep the hand-scheduled kernels")
                assert p["slots"] % cus == 0 and p["tiles"] == p["tiles_m"] * p["tiles_n"]
                if p["plan"] == "plain":
                    assert p["wgs"] == p["tiles"] and p["slices"] == 1
                elif p["plan"] == "strided":          # one persistent workgroup per slot, more tiles than slots, whole tiles only
                    assert p["wgs"] == p["slots"] < p["tiles"] and p["slices"] == 1
                elif p["plan"] == "hybrid":           # (one-chain mode only) whole rounds strided, >= 8 remaining tiles cut along K
                    assert not laser and p["wgs"] == p["slots"] < p["tiles"] and p["tiles"] % p["wgs"] >= 8 and p["slices"] >= 1
                else:                                  # K-slice cuts: at most every slot of THIS device, a multiple of 8 when tiles are cut
                    assert 8 <= p["wgs"] <= p["slots"] and p["slices"] >= 1
        # the number of rounds the busiest CU works follows the CU count: fewer CUs -> the same shape needs a plan with fewer workgroups
        assert la.plan_f32(8192, 8192, 8192, True, cus)["wgs"] == cus
    # problems of a few tiles keep the compiler-scheduled small / slice-parallel forms on the full chip, and reach the tiled kernels on
    # a partition where they are several rounds of work
    assert la.plan_f32(512, 512, 512, True, 256)["kernel"] == 0
    assert la.plan_f32(512, 512, 512, True, 32)["kernel"] != 0
    with pytest.raises(la.LaserHipError):
        la.plan_f32(0, 4, 4)
