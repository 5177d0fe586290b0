"""GPU tests of the assembly kernels' launch plans (laser_amd/csrc/gemm_f32_asm.cpp plan_launch, asmgen/f32_kernel.py sched_next):
the workgroup -> tile map is arithmetic in the kernel (no table, no lock, nothing to upload), launches may be persistent (fewer
workgroups than tiles) and may cut tiles at K-slice boundaries, the owner of a tile's slice 0 handing its running sum on to the
workgroup that owns the rest.  Laser-order results must be the SAME bits whatever the plan (gemm.nim:150-158: slices are independent chains,
their sums are added in order) and equal to the oracle; one-chain results stay within 1e-5 mean relative error
(gemm_bench_float32.nim:365-367).  Also: concurrent launches from several host threads and streams, and a launch captured into a
HIP graph and replayed."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import laser_amd
    laser_amd.lib()
    laser_amd.set_float_mode(0)
    laser_amd.set_f32_config(-1)
    laser_amd.set_option("slice_parallel", 0)      # (few-tile x long-K problems: keep them on the tiled launcher under test)
    yield laser_amd
    laser_amd.set_option("slice_parallel", 1)
    for k, v in (("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("asm_slice", 0), ("f32_asm", 1), ("f64_asm", 1)):
        laser_amd.set_option(k, v)
    laser_amd.set_float_mode(0)


def _rnd(rng, shape, dtype=np.float32):
    return rng.uniform(-0.1, 0.1, shape).astype(dtype)


def _mre(got, want):
    return float(np.mean(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)))


# kernel index (gemm_f32_asm.cpp kKernels) -> tile; laser-order kernels only here
LASER_KERNELS = {0: (256, 128), 2: (128, 128), 12: (64, 64)}
LASER_KERNELS_NT = {4: (256, 128), 6: (128, 128), 14: (64, 64)}


@pytest.mark.parametrize("nt", [False, True])
def test_laser_order_bit_identical_for_every_plan_and_workgroup_count(la, oracle, nt):
    """every laser-order kernel, forced; plain launch vs persistent launches of 7 .. 768 workgroups (cuts in every position: ranges
    shorter than a tile, longer than a tile, ending inside the first / last slice), ragged M / N / K, alpha / beta"""
    import torch
    rng = np.random.default_rng(4242 + nt)
    la.set_option("f32_asm", 2)
    try:
        for kern, (bm, bn) in (LASER_KERNELS_NT if nt else LASER_KERNELS).items():
            M, N, K = 3 * bm + 17, 5 * bn - 9, 2048 + 100 + 3
            A = torch.from_numpy(_rnd(rng, (M, K))).cuda()
            Bh = _rnd(rng, (K, N))
            B = torch.from_numpy(np.ascontiguousarray(Bh.T)).cuda().t() if nt else torch.from_numpy(Bh).cuda()
            C0 = torch.from_numpy(_rnd(rng, (M, N))).cuda()
            want = oracle.matmul(A.cpu().numpy(), Bh, 0.5, -0.25, C0.cpu().numpy())
            la.set_option("asm_kernel", kern)
            for plan, wgs, late in ((1, 0, 0), (2, 0, 0), (2, 7, 0), (2, 7, 1), (2, 15, 1), (2, 16, 0), (2, 100, 0), (2, 100, 1), (2, 256, 0), (2, 333, 1)):
                la.set_option("asm_plan", plan)
                la.set_option("asm_wgs", wgs)
                la.set_option("asm_noseed", late)        # 1: a piece never takes its received sum early (the two-run receive path)
                C = C0.clone()
                la.matmul(A, B, 0.5, -0.25, C)
                assert la.last_f32_asm() == kern + 1, (kern, plan, wgs, la.last_f32_asm())
                if plan == 2:
                    assert la.get_option("last_asm_slices") == 5, (kern, la.get_option("last_asm_slices"))
                    if wgs:
                        units = 5 * ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
                        want_wgs = min(wgs, units)
                        if want_wgs >= 8:      # 8 XCDs x G / 8 workgroups, no more per XCD than its units
                            want_wgs = 8 * min(want_wgs // 8, (units // 5 // 8) * 5)
                        assert la.get_option("last_asm_wgs") == want_wgs
                assert np.array_equal(C.cpu().numpy(), want), (kern, plan, wgs)
    finally:
        for k, v in (("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("f32_asm", 1), ("asm_noseed", 0)):
            la.set_option(k, v)


def test_hybrid_plan_bit_identical(la, oracle):
    """Round 6: the two-launch plan (asm_plan = 4) -- the whole rounds of the raster as a strided launch with pipelined transitions, the
    remaining tiles as a second launch of the same kernel that cuts them along K (the kernels' tile base) -- against the oracle bit for
    bit in laser-order mode (the hand-over folds the kc slices in order: same bits under every plan), within the one-chain tolerance
    in fast mode; workgroup counts that leave 8 .. 30 tiles to the second launch, the early and the late receive path."""
    import torch
    rng = np.random.default_rng(414)
    la.set_option("f32_asm", 2)
    try:
        for mode in (0, 1):
            la.set_float_mode(mode)
            for kern, (bm, bn), nt in (((0, 8)[mode], (256, 128), False), ((12, 13)[mode], (64, 64), False), ((30, 31)[mode], (128, 128), False),
                                       ((46, 47)[mode], (96, 96), False), ((6, 7)[mode], (128, 128), True), ((62, 63)[mode], (160, 160), False)):
                M, N, K = 10 * bm - 17, 13 * bn - 5, 1600               # 130 tiles, 3.1 kc slices
                Ah, Bh = _rnd(rng, (M, K)), _rnd(rng, (K, N))
                A = torch.from_numpy(Ah).cuda()
                B = torch.from_numpy(np.ascontiguousarray(Bh.T)).cuda().t() if nt else torch.from_numpy(Bh).cuda()
                want = oracle.matmul(Ah, Bh) if mode == 0 else Ah.astype(np.float64) @ Bh.astype(np.float64)
                la.set_option("asm_kernel", kern)
                for wgs, late in ((24, 0), (24, 1), (40, 0), (61, 1), (100, 0)):
                    la.set_option("asm_plan", 4)
                    la.set_option("asm_wgs", wgs)
                    la.set_option("asm_noseed", late)
                    C = torch.full((M, N), float("nan"), device="cuda")
                    la.matmul(A, B, 1, 0, C)
                    assert la.last_f32_asm() == kern + 1, (kern, wgs, la.last_f32_asm())
                    assert la.get_option("last_asm_wgs") == wgs and la.get_option("last_asm_rem") == 130 % wgs, (kern, wgs, la.get_option("last_asm_rem"))
                    got = C.cpu().numpy()
                    if mode == 0:
                        assert np.array_equal(got, want), (kern, wgs, late)
                    else:
                        assert _mre(got, want) <= 1e-5, (kern, wgs, _mre(got, want))
    finally:
        la.set_float_mode(0)
        for k, v in (("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("f32_asm", 1), ("asm_noseed", 0)):
            la.set_option(k, v)


def test_one_chain_cut_launches_within_tolerance_and_deterministic(la, oracle):
    """FAST mode: a cut launch adds partial sums (another rounding order than the single chain, by design): <= 1e-5 mean relative
    error against the float64 product, and the same bits on every repetition (the fix-up order is fixed, not first-come)"""
    import torch
    rng = np.random.default_rng(99)
    la.set_float_mode(1)
    la.set_option("f32_asm", 2)
    try:
        for kern, (bm, bn) in {1: (256, 256), 8: (256, 128), 3: (128, 128), 13: (64, 64)}.items():
            M, N, K = 2 * bm + 40, 3 * bn - 8, 1500
            Ah, Bh = _rnd(rng, (M, K)), _rnd(rng, (K, N))
            A, B = torch.from_numpy(Ah).cuda(), torch.from_numpy(Bh).cuda()
            want = Ah.astype(np.float64) @ Bh.astype(np.float64)
            la.set_option("asm_kernel", kern)
            for wgs, sl in ((0, 0), (11, 4), (50, 7), (256, 0)):
                la.set_option("asm_plan", 2)
                la.set_option("asm_wgs", wgs)
                la.set_option("asm_slice", sl)
                outs = []
                for _ in range(2):
                    C = torch.full((M, N), float("nan"), device="cuda")
                    la.matmul(A, B, 1, 0, C)
                    assert la.last_f32_asm() == kern + 1, (kern, la.last_f32_asm())
                    outs.append(C.cpu().numpy())
                assert np.array_equal(outs[0], outs[1]), (kern, wgs, sl)
                assert _mre(outs[0], want) <= 1e-5, (kern, wgs, sl, _mre(outs[0], want))
    finally:
        la.set_float_mode(0)
        for k, v in (("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("asm_slice", 0), ("f32_asm", 1)):
            la.set_option(k, v)


def test_f64_plans_bit_identical(la, oracle):
    """float64 kernels (kc = 256): every plan against the oracle, laser-order"""
    import torch
    rng = np.random.default_rng(5)
    la.set_option("f64_asm", 2)
    try:
        for kern, (bm, bn), nt in ((16, (128, 128), False), (18, (64, 64), False), (27, (64, 64), True)):
            M, N, K = 3 * bm + 10, 2 * bn + 30, 1024 + 130
            Ah, Bh = _rnd(rng, (M, K), np.float64), _rnd(rng, (K, N), np.float64)
            A = torch.from_numpy(Ah).cuda()
            B = torch.from_numpy(np.ascontiguousarray(Bh.T)).cuda().t() if nt else torch.from_numpy(Bh).cuda()
            C0 = torch.from_numpy(_rnd(rng, (M, N), np.float64)).cuda()
            want = oracle.matmul(Ah, Bh, -1.5, 0.5, C0.cpu().numpy())
            la.set_option("asm_kernel", kern)
            for plan, wgs, late in ((1, 0, 0), (2, 0, 0), (2, 9, 1), (2, 64, 0), (2, 64, 1), (2, 256, 0)):
                la.set_option("asm_plan", plan)
                la.set_option("asm_wgs", wgs)
                la.set_option("asm_noseed", late)
                C = C0.clone()
                la.matmul(A, B, -1.5, 0.5, C)
                assert la.get_option("last_f64_asm") == kern + 1, (kern, plan, wgs, la.get_option("last_f64_asm"))
                assert np.array_equal(C.cpu().numpy(), want), (kern, plan, wgs)
            # round 6: the strided plan with pipelined transitions (beta == 0, K a multiple of 16): workgroup counts that give every
            # workgroup several tiles, against the plain launch's bits and the oracle
            K2 = 1024 + 128
            A2, B2 = A[:, :K2].contiguous(), (B[:K2, :] if not nt else torch.from_numpy(np.ascontiguousarray(Bh[:K2].T)).cuda().t())
            want2 = oracle.matmul(Ah[:, :K2], Bh[:K2], 1.0, 0.0, None)
            la.set_option("asm_noseed", 0)
            for plan, wgs in ((1, 0), (3, 0), (3, 2), (3, 5), (3, 11)):
                la.set_option("asm_plan", plan)
                la.set_option("asm_wgs", wgs)
                C = torch.full((M, N), float("nan"), dtype=torch.float64, device="cuda")
                la.matmul(A2, B2, 1.0, 0.0, C)
                assert la.get_option("last_f64_asm") == kern + 1, (kern, plan, wgs, la.get_option("last_f64_asm"))
                if plan == 3 and wgs:
                    assert la.get_option("last_asm_wgs") == wgs, (kern, wgs, la.get_option("last_asm_wgs"))
                assert np.array_equal(C.cpu().numpy(), want2), (kern, plan, wgs)
    finally:
        for k, v in (("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("f64_asm", 1), ("asm_noseed", 0)):
            la.set_option(k, v)


def test_default_plan_mid_sizes_bit_exact(la, oracle):
    """the shapes the persistent plan was built for, default options: whichever plan the model takes, the oracle's bits"""
    import torch
    rng = np.random.default_rng(17)
    for (M, N, K) in ((1920, 1920, 1920), (1536, 1536, 1536), (1024, 1024, 1024), (1000, 3000, 2000), (3072, 3072, 1030)):
        Ah, Bh = _rnd(rng, (M, K)), _rnd(rng, (K, N))
        C = la.matmul(torch.from_numpy(Ah).cuda(), torch.from_numpy(Bh).cuda())
        assert la.last_f32_asm() != 0, (M, N, K)
        assert np.array_equal(C.cpu().numpy(), oracle.matmul(Ah, Bh)), (M, N, K, la.last_f32_asm(), la.get_option("last_asm_wgs"))


def test_concurrent_launches_two_threads_two_streams(la, oracle):
    """distinct shapes launched at the same time from 2 host threads x 2 streams each (plain and cut plans mixed): the launcher
    holds no lock across a launch, keeps one workspace per stream, and has nothing to upload -- every result bit-exact"""
    import torch
    rng = np.random.default_rng(2024)
    shapes = [(1100, 900, 1600), (700, 1300, 2100), (1536, 1536, 1100), (520, 2050, 1030)]
    probs = []
    for (M, N, K) in shapes:
        Ah, Bh = _rnd(rng, (M, K)), _rnd(rng, (K, N))
        probs.append((torch.from_numpy(Ah).cuda(), torch.from_numpy(Bh).cuda(), oracle.matmul(Ah, Bh)))
    torch.cuda.synchronize()
    la.set_option("f32_asm", 2)
    errors = []

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            outs = []
            for rep in range(6):
                for si, st in enumerate(streams):
                    A, B, want = probs[(2 * tid + si + rep) % len(probs)]
                    with torch.cuda.stream(st):
                        la.set_option("asm_plan", 2 if (rep + si) % 2 else 0)     # (a process-wide knob: any mix is legal)
                        outs.append((la.matmul(A, B), want))
            for st in streams:
                st.synchronize()
            for C, want in outs:
                if not np.array_equal(C.cpu().numpy(), want):
                    errors.append(f"thread {tid}: mismatch")
        except Exception as exc:      # noqa: BLE001
            errors.append(f"thread {tid}: {type(exc).__name__}: {exc}")

    try:
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        la.set_option("asm_plan", 0)
        la.set_option("f32_asm", 1)
    assert not errors, errors


def test_dev_entry_point_captured_into_a_graph_and_replayed(la, oracle):
    """a _dev call is pure stream work: captured into a HIP graph (no allocation, no copy, no synchronisation inside the call) and
    replayed with new operand contents -- both the plain plan and, once the stream's workspace exists, a cut plan"""
    import torch
    rng = np.random.default_rng(31)
    M, N, K = 1536, 1536, 1100
    A = torch.from_numpy(_rnd(rng, (M, K))).cuda()
    B = torch.from_numpy(_rnd(rng, (K, N))).cuda()
    C = torch.zeros((M, N), device="cuda")
    la.set_option("f32_asm", 2)
    try:
        for plan in (1, 2, 3, 4):             # (3, 4: the strided plan and the two-launch hybrid; 40 workgroups so that both have something to do)
            la.set_option("asm_plan", plan)
            la.set_option("asm_wgs", 40 if plan >= 3 else 0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                la.matmul(A, B, 1, 0, C)          # warm: module load, (plan 2) this stream's workspace
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                la.matmul(A, B, 1, 0, C)
            assert la.last_f32_asm() != 0
            for rep in range(3):
                Ah = _rnd(rng, (M, K))
                A.copy_(torch.from_numpy(Ah))
                C.fill_(float("nan"))
                g.replay()
                torch.cuda.synchronize()
                assert np.array_equal(C.cpu().numpy(), oracle.matmul(Ah, B.cpu().numpy())), (plan, rep)
        # a convolution whose main launch walks units (no workspace, nothing allocated): captured and replayed the same way
        la.set_option("asm_plan", 0)
        la.set_option("asm_wgs", 0)
        la.set_option("conv_walk", 5)
        ishape, kshape = (4, 64, 30, 30), (256, 64, 3, 3)
        x = torch.from_numpy(_rnd(rng, ishape)).cuda()
        w = torch.from_numpy(_rnd(rng, kshape)).cuda()
        oshape = la.conv2d_out_shape(ishape, kshape, (1, 1), (1, 1))
        o = torch.zeros(oshape, device="cuda")
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            la.conv2d_im2col(o, oshape, x, ishape, w, kshape, (1, 1), (1, 1), None)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            la.conv2d_im2col(o, oshape, x, ishape, w, kshape, (1, 1), (1, 1), None)
        assert la.last_f32_asm() >= 67
        for rep in range(2):
            xh = _rnd(rng, ishape)
            x.copy_(torch.from_numpy(xh))
            o.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(o.cpu().numpy(), oracle.conv2d_im2col(xh, w.cpu().numpy(), (1, 1), (1, 1), isa=oracle.fused_isa(np.float32))), rep
    finally:
        la.set_option("asm_plan", 0)
        la.set_option("asm_wgs", 0)
        la.set_option("conv_walk", 1)
        la.set_option("f32_asm", 1)


def test_full_size_c3_transposed_b_dense_c_every_element(la, oracle):
    """BASELINE configs[2] in the exact operand form bench.py times: A dense, B passed transposed (rowStrideB = 1), C dense,
    4096^3 -> the `_nt` assembly kernels; every element against the oracle"""
    import torch
    rng = np.random.default_rng(6)
    n = 4096
    Ah, Bt = _rnd(rng, (n, n)), _rnd(rng, (n, n))           # Bt = B^T stored row-major
    A, B = torch.from_numpy(Ah).cuda(), torch.from_numpy(Bt).cuda().t()
    C = torch.zeros((n, n), device="cuda")
    la.matmul(A, B, 1, 0, C)
    assert la.last_f32_asm() in (5, 7, 15, 33, 49, 53, 57, 61, 65), la.last_f32_asm()
    assert np.array_equal(C.cpu().numpy(), oracle.matmul(Ah, np.ascontiguousarray(Bt.T)))


def test_fused_prologue_relu_on_a_and_b(la, oracle):
    """README.md:243-244 (the reference's roadmap): an operation fused BEFORE the product -- relu(A) and / or relu(B), applied in the
    staging registers of the `_pre` assembly kernels (kernel indices 35..46), in the packing pass of a strided operand, or
    materialised for paths without a kernel (float64) -- C = act(alpha relu(A) relu(B) + beta C + bias), bit-exact against the
    oracle run on operands with relu applied on the host, both accumulation orders' kernels, B plain and transposed."""
    import torch
    rng = np.random.default_rng(321)
    relu = lambda x: np.where(x > 0, x, 0).astype(x.dtype)
    for (M, N, K), nt in (((1100, 1300, 1500), False), ((2048, 2048, 1030), True), ((520, 700, 600), False)):
        Ah, Bh = _rnd(rng, (M, K)), _rnd(rng, (K, N))
        A = torch.from_numpy(Ah).cuda()
        B = torch.from_numpy(np.ascontiguousarray(Bh.T)).cuda().t() if nt else torch.from_numpy(Bh).cuda()
        C0 = _rnd(rng, (M, N))
        bias = _rnd(rng, (N,))
        for pre in (la.PRE_RELU_A, la.PRE_RELU_B, la.PRE_RELU_A | la.PRE_RELU_B):
            for mode in (0, 1):
                beta = 0.25 if mode == 0 else 0.0      # (the one-chain kernels' fused epilogue has no C read)
                la.set_float_mode(mode)
                try:
                    C = torch.from_numpy(C0.copy()).cuda()
                    la.matmul(A, B, 0.5, beta, C, bias=torch.from_numpy(bias).cuda(), activation="relu", pre=pre)
                    used = la.last_f32_asm()
                finally:
                    la.set_float_mode(0)
                assert 35 <= used <= 46, (M, N, K, pre, mode, used)
                Ar, Br = (relu(Ah) if pre & la.PRE_RELU_A else Ah), (relu(Bh) if pre & la.PRE_RELU_B else Bh)
                if mode == 0:
                    want = relu((oracle.matmul(Ar, Br, 0.5, beta, C0.copy()) + bias[None, :]).astype(np.float32))
                    assert np.array_equal(C.cpu().numpy(), want), (M, N, K, pre)
                else:
                    want = np.maximum(0.5 * (Ar.astype(np.float64) @ Br.astype(np.float64)) + bias[None, :], 0)
                    assert np.allclose(C.cpu().numpy(), want, rtol=1e-4, atol=1e-5), (M, N, K, pre)
    # a strided A (every second column): the prologue happens in the packing pass; float64: materialised
    Ah2 = _rnd(rng, (1100, 3000))
    A2 = torch.from_numpy(Ah2).cuda()[:, ::2]
    Bh = _rnd(rng, (1500, 1300))
    C = la.matmul(A2, torch.from_numpy(Bh).cuda(), pre=la.PRE_RELU_A)
    assert la.last_f32_asm() != 0
    assert np.array_equal(C.cpu().numpy(), oracle.matmul(relu(np.ascontiguousarray(Ah2[:, ::2])), Bh))
    Ad, Bd = _rnd(rng, (300, 400), np.float64), _rnd(rng, (400, 500), np.float64)
    Cd = la.matmul(torch.from_numpy(Ad).cuda(), torch.from_numpy(Bd).cuda(), pre=la.PRE_RELU_A | la.PRE_RELU_B)
    assert np.array_equal(Cd.cpu().numpy(), oracle.matmul(relu(Ad), relu(Bd)))
    # host-pointer entry point
    Ch = la.matmul(Ah2[:, :1500].copy(), Bh, pre=la.PRE_RELU_B)
    assert np.array_equal(Ch, oracle.matmul(Ah2[:, :1500].copy(), relu(Bh)))


def test_no_fix_up_ever_gave_up_waiting(la):
    """(last in this file) the error word in front of every stream's flags: a fix-up that polled for ~2 s without seeing its partial
    would have counted itself here -- a lost release or a scheduling order the design does not allow"""
    assert la.get_option("asm_fixup_timeouts") == 0


@pytest.mark.parametrize("mode", [0, 1])
def test_strided_persistent_plan_with_pipelined_tile_transitions(la, oracle, mode):
    """Round 6 (asmgen/f32_kernel.py Cfg.pipe, gemm_f32_asm.cpp plan_launch): with more tiles than workgroup slots the 256x128 kernels
    run as ONE workgroup per CU that walks tiles v, v + 256, ... without leaving its K loop -- the last bodies of a tile fetch the
    next tile's first K-tiles, the next tile's first body stores this tile's C.  Same bits as one workgroup per tile (also in
    one-chain mode: the chain order does not change), equal to the oracle in laser-order mode; ragged edges, a strided C view, every
    workgroup count (asm_wgs), and the cases that must NOT pipeline (beta, K tail) on the same plan."""
    import torch
    rng = np.random.default_rng(606 + mode)
    la.set_option("f32_asm", 2)
    la.set_float_mode(mode)
    try:
        for (M, N, K, kw) in [(4096, 4096, 1024, {}), (4100, 4228, 544, {}), (3000, 5000, 96, dict(csc=2)), (2560, 4100, 1056, dict(alpha=0.5)),
                              (2048, 4096, 640, dict(beta=0.25)), (2304, 4000, 1000, {})]:
            alpha, beta, csc = kw.get("alpha", 1.0), kw.get("beta", 0.0), kw.get("csc", 1)
            A = torch.from_numpy(_rnd(rng, (M, K))).cuda()
            B = torch.from_numpy(_rnd(rng, (K, N))).cuda()
            C0 = torch.from_numpy(_rnd(rng, (M, N * csc))).cuda()
            outs = {}
            ragged = M % 256 != 0 or N % 128 != 0       # (the caller may cut a ragged problem into a main launch of whole tiles + edge launches)
            # (one-chain mode: the model's own choice may cut K -- another, equally valid rounding -- so only the plans that never do are compared)
            for plan, wgs in ((1, 0), (0, 0), (3, 0), (3, 100), (3, 37)) if mode == 0 else ((1, 0), (3, 0), (3, 100), (3, 37)):
                kern = 8 if (mode and K > 512) else 0       # (K <= kc is one chain in both modes: the laser-order kernel serves it)
                la.set_option("asm_kernel", kern)
                la.set_option("asm_plan", plan)
                la.set_option("asm_wgs", wgs)
                C = C0.clone()
                la.matmul(A, B, alpha, beta, C[:, ::csc])
                assert la.last_f32_asm() == kern + 1, (M, N, K, plan, la.last_f32_asm())
                g = la.get_option("last_asm_wgs")
                tiles = -(-M // 256) * -(-N // 128)
                may = beta == 0 and K % 32 == 0 and K >= 96      # (else the launcher keeps one workgroup per tile: nothing to pipeline)
                if ragged:
                    assert g <= tiles
                elif plan == 1 or (plan == 3 and not may):
                    assert g == tiles
                elif plan == 3:
                    assert g == min(wgs or 256, tiles), (plan, wgs, g)
                elif may:
                    assert g == 256, "the launch model takes the strided plan when there are more tiles than workgroup slots"
                outs[(plan, wgs)] = C
            ref = outs[(1, 0)]
            for key, C in outs.items():
                assert torch.equal(C, ref), (M, N, K, kw, key)
            if csc > 1:
                assert torch.equal(ref[:, 1::csc], C0[:, 1::csc]), "wrote between the columns of the view"
            if mode == 0:
                want = oracle.matmul(A.cpu().numpy(), B.cpu().numpy(), alpha, beta, C0.cpu().numpy()[:, ::csc])
                assert np.array_equal(ref.cpu().numpy()[:, ::csc], want), (M, N, K, kw)
    finally:
        for k, v in (("asm_plan", 0), ("asm_kernel", -1), ("asm_wgs", 0), ("f32_asm", 1)):
            la.set_option(k, v)
        la.set_float_mode(0)


def test_a_hand_over_that_gives_up_is_reported_on_the_next_call(la):
    """VERDICT r5 missing #3: a receiver of a cut launch that gives up waiting for its predecessor's running sum used to go on with
    whatever the slot held while the API call returned 0.  Now every such workgroup counts itself in the stream's error word (an atomic
    add), the word travels back behind the launch (an asynchronous 4-byte copy, no synchronisation), and the NEXT call on that stream
    fails with LASER_HIP_E_HIP and a message naming the stream; the stream is usable again after the report.  Option
    "asm_test_giveup" makes every receiver give up at once (the hand-over data is there in this test: only the report is exercised)."""
    import torch
    rng = np.random.default_rng(77)
    M, N, K = 1536, 1536, 4096          # 288 tiles of 128x128... the cut plan shares K-slice units: some tiles are handed over
    A = torch.from_numpy(_rnd(rng, (M, K))).cuda()
    B = torch.from_numpy(_rnd(rng, (K, N))).cuda()
    C = torch.zeros((M, N), device="cuda")
    la.set_option("f32_asm", 2)
    try:
        la.set_option("asm_plan", 2)
        la.set_option("asm_kernel", 30)               # 144 tiles of 128x128 x 8 slices over 256 workgroups = 4.5 units each: tiles ARE handed over
        la.matmul(A, B, 1, 0, C)                      # (the model's own pick here -- 256 tiles of 96x96, one per workgroup -- cuts nothing)
        assert la.get_option("last_asm_slices") > 1 and la.last_f32_asm() == 31, "this shape must run as a K-cut launch"
        torch.cuda.synchronize()
        ref = C.clone()
        la.set_option("asm_test_giveup", 1)
        la.matmul(A, B, 1, 0, C)                      # the launch whose receivers give up: returns 0 (the report is asynchronous)
        la.set_option("asm_test_giveup", 0)
        torch.cuda.synchronize()
        with pytest.raises(la.LaserHipError) as e:    # the next call on the stream is refused and says why
            la.matmul(A, B, 1, 0, C)
        assert "gave up waiting" in str(e.value) and "stream" in str(e.value), str(e.value)
        la.matmul(A, B, 1, 0, C)                      # reported once; the flags were reset: clean again
        torch.cuda.synchronize()
        assert torch.equal(C, ref)
        assert la.get_option("asm_fixup_timeouts") == 0
    finally:
        for k, v in (("asm_plan", 0), ("asm_test_giveup", 0), ("f32_asm", 1), ("asm_kernel", -1)):
            la.set_option(k, v)
