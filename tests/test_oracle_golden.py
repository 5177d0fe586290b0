"""Pin the CPU oracle (oracle/) before trusting it.

1. every known-answer test the reference holds for this path (tests/golden/laser_kats.json,
   extracted by tests/golden/make_kats.py from gemm.nim:255-507, gemm_prepacked.nim:352-523,
   conv2d_common.nim:139-283), for every dtype and every restated ISA geometry;
2. independent cross-checks: float64 naive product, numpy/OpenBLAS (the reference's own
   "vendor BLAS" comparator, bar <= 1e-5 mean relative error, gemm_bench_float32.nim:365-367),
   torch conv2d, the direct convolution;
3. invariants of the restatement: SIMD micro-kernels == scalar micro-kernel bitwise; result
   independent of MR/NR geometry and of thread count; fp32 result == explicit per-element
   "fmaf chain per kc=512 slice, slices added in order" model (SURVEY.md section 8a contract).
"""
import numpy as np
import pytest

DTYPES = [np.float32, np.float64, np.int32, np.int64]
ALL_ISA = list(range(8))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("isa", ALL_ISA)
def test_gemm_kats_every_dtype_every_isa(oracle, kats, dtype, isa):
    for k in kats["gemm"]:
        A = np.array(k["A"], dtype=dtype)
        B = np.array(k["B"], dtype=dtype)
        C = np.full((k["M"], k["N"]), 77, dtype=dtype)  # beta = 0 must overwrite
        oracle.matmul(A, B, 1, 0, C, isa=isa)
        assert np.array_equal(C, np.array(k["C"], dtype=dtype)), k["source"]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("isa", [0, 5, 7])
def test_prepacked_kats(oracle, kats, dtype, isa):
    for k in kats["gemm_prepacked"]:
        A = np.array(k["A"], dtype=dtype)
        B = np.array(k["B"], dtype=dtype)
        C = oracle.gemm_prepack_and_run(A, B, isa=isa)
        assert np.array_equal(C, np.array(k["C"], dtype=dtype)), k["source"]


def test_conv_kats(oracle, kats):
    for c in kats["conv"]:
        x = np.array(c["input"], dtype=np.float32).reshape(c["ishape"])
        w = np.array(c["kernel"], dtype=np.float32).reshape(c["kshape"])
        y = oracle.conv2d_im2col(x, w, c["padding"], c["strides"])
        assert y.ravel().tolist() == c["target"], c["source"]
        yd = oracle.conv2d_direct(x, w, c["padding"], c["strides"])
        assert yd.ravel().tolist() == c["target"], c["source"]


def _fma_chain_model(A, B, alpha, beta, C0, kc=512):
    """Per-element statement of the accumulation-order contract, in numpy float32 + math.fma-free
    emulation: fma(a,b,c) == float32(float64(a)*float64(b) + float64(c)) is exact for f32 inputs
    (the double product is exact and the double sum rounds once more only when it is not
    representable -- double rounding cannot change an f32 fma result because 2*24+... < 53 holds
    for the product; the sum can double-round in rare ties, so we use integer-valued data here)."""
    M, K = A.shape
    N = B.shape[1]
    C = np.array(C0, dtype=np.float32, copy=True)
    first = True
    for pc in range(0, K, kc):
        S = np.zeros((M, N), dtype=np.float32)
        for k in range(pc, min(K, pc + kc)):
            S = (A[:, k:k + 1].astype(np.float64) * B[k:k + 1, :].astype(np.float64) + S.astype(np.float64)).astype(np.float32)
        b = beta if first else np.float32(1)
        if b == 0:
            C = np.zeros_like(C)
        elif b != 1:
            C = (C * np.float32(b)).astype(np.float32)
        C = (C + (S if alpha == 1 else (np.float32(alpha) * S).astype(np.float32))).astype(np.float32)
        first = False
    return C


def test_f32_accumulation_order_contract(oracle):
    rng = np.random.default_rng(1)
    M, N, K = 37, 45, 1100  # 3 kc slices, ragged tiles for every geometry
    A = rng.integers(-8, 9, (M, K)).astype(np.float32) / 8
    B = rng.integers(-8, 9, (K, N)).astype(np.float32) / 8
    C0 = rng.integers(-8, 9, (M, N)).astype(np.float32)
    for alpha, beta in [(1, 0), (1, 1), (0.5, 0.25), (2, 0)]:
        want = _fma_chain_model(A, B, np.float32(alpha), np.float32(beta), C0)
        for isa in (5, 7):
            got = oracle.matmul(A, B, alpha, beta, C0.copy(), isa=isa)
            assert np.array_equal(got, want), (alpha, beta, isa)


@pytest.mark.parametrize("dtype", DTYPES)
def test_simd_equals_scalar_and_geometry_invariance(oracle, dtype):
    rng = np.random.default_rng(2)
    M, N, K = 203, 131, 777
    if np.dtype(dtype).kind == "f":
        A = rng.uniform(-0.1, 0.1, (M, K)).astype(dtype)
        B = rng.uniform(-0.1, 0.1, (K, N)).astype(dtype)
    else:
        info = np.iinfo(dtype)
        A = rng.integers(info.min, info.max, (M, K), dtype=dtype)  # full range: wrap-around
        B = rng.integers(info.min, info.max, (K, N), dtype=dtype)
    ref = oracle.matmul(A, B, isa=7, use_simd=False)
    assert np.array_equal(oracle.matmul(A, B, isa=7, use_simd=True), ref)
    fused_other = 5 if np.dtype(dtype).kind == "f" else 6
    assert np.array_equal(oracle.matmul(A, B, isa=fused_other, use_simd=True), ref)
    assert np.array_equal(oracle.matmul(A, B, isa=fused_other, use_simd=False), ref)
    if np.dtype(dtype).kind != "f":  # integers: any order is bit-exact
        want = (A.astype(object) @ B.astype(object))
        bits = 8 * np.dtype(dtype).itemsize
        want = np.vectorize(lambda v: ((int(v) + (1 << (bits - 1))) % (1 << bits)) - (1 << (bits - 1)))(want).astype(dtype)
        assert np.array_equal(ref, want)


def test_thread_count_independence(oracle):
    rng = np.random.default_rng(3)
    M, N, K = 400, 300, 600  # > 128^3 -> parallel path
    A = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
    B = rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32)
    n0 = oracle.num_threads()
    try:
        oracle.set_num_threads(1)
        r1 = oracle.matmul(A, B)
        oracle.set_num_threads(max(2, n0))
        rn = oracle.matmul(A, B)
    finally:
        oracle.set_num_threads(n0)
    assert np.array_equal(r1, rn)


def test_vs_float64_and_openblas(oracle):
    rng = np.random.default_rng(42)
    M = N = K = 384
    A = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
    B = rng.uniform(-0.1, 0.1, (K, N)).astype(np.float32)
    got = oracle.matmul(A, B)
    f64 = oracle.naive_gemm_f64(A, B)
    assert np.allclose(f64, A.astype(np.float64) @ B.astype(np.float64), rtol=1e-12, atol=1e-12)
    assert oracle.mean_relative_error(got, f64.astype(np.float32)) <= 1e-5
    assert oracle.mean_relative_error(got, A @ B) <= 1e-5  # numpy == OpenBLAS 0.3.x
    assert np.max(np.abs(got - f64)) < 1e-5


def test_strides_and_semantics(oracle):
    rng = np.random.default_rng(5)
    M, N, K = 70, 50, 90
    Abig = rng.uniform(-1, 1, (2 * M, K)).astype(np.float32)
    Bt = rng.uniform(-1, 1, (N, K)).astype(np.float32)  # stored transposed
    A = Abig[::2]          # every-2nd-row view (README.md:211-213)
    B = Bt.T               # (rowStride=1, colStride=K)
    Cbuf = np.full((M, 2 * N), np.nan, dtype=np.float32)
    Cv = Cbuf[:, ::2]      # colStrideC = 2
    oracle.matmul(A, B, 1, 0, Cv)
    want = oracle.matmul(np.ascontiguousarray(A), np.ascontiguousarray(B))
    assert np.array_equal(Cv, want)
    assert np.isnan(Cbuf[:, 1::2]).all()  # untouched gaps
    # beta == 0 never reads C (NaN-safe); K == 0 leaves C untouched even if beta != 1
    C = np.full((M, N), np.nan, dtype=np.float32)
    assert not np.isnan(oracle.matmul(np.ascontiguousarray(A), np.ascontiguousarray(B), 1, 0, C)).any()
    C = np.full((4, 4), 3.0, dtype=np.float32)
    oracle.matmul(np.zeros((4, 0), np.float32), np.zeros((0, 4), np.float32), 1, 0.5, C)
    assert (C == 3.0).all()


def test_transposes(oracle):
    rng = np.random.default_rng(6)
    for shape in [(1, 1), (33, 65), (100, 7), (4000 // 8, 2000 // 8)]:
        x = rng.standard_normal(shape).astype(np.float32)
        assert np.array_equal(oracle.transpose2D_copy(x), x.T)
    x = rng.integers(-100, 100, (3, 45, 70)).astype(np.int32)
    assert np.array_equal(oracle.transpose2D_batched(x), x.transpose(0, 2, 1))
    x = rng.standard_normal((2, 5, 6, 7))
    assert np.array_equal(oracle.nchw2nhwc(x), x.transpose(0, 2, 3, 1))
    assert np.array_equal(oracle.nhwc2nchw(oracle.nchw2nhwc(x)), x)


def test_conv_vs_torch(oracle):
    import torch
    rng = np.random.default_rng(7)
    for (ishape, kshape, pad, st) in [((2, 5, 13, 11), (4, 5, 3, 3), (1, 1), (1, 1)),
                                      ((1, 3, 9, 14), (2, 3, 3, 2), (0, 1), (2, 1)),
                                      ((2, 8, 6, 6), (5, 8, 1, 1), (0, 0), (1, 1))]:
        x = rng.uniform(0, 1, ishape).astype(np.float32)
        w = rng.uniform(0, 1, kshape).astype(np.float32)
        y = oracle.conv2d_im2col(x, w, pad, st)
        t = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=pad, stride=st).numpy()
        assert y.shape == t.shape
        assert oracle.mean_relative_error(y, t) <= 1e-5
        # im2col itself == unfold
        unf = torch.nn.functional.unfold(torch.from_numpy(x[:1]), kshape[2:], padding=pad, stride=st)[0].numpy()
        assert np.array_equal(oracle.im2col(x[0], kshape, pad, st), unf)
