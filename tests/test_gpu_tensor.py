"""Device-resident Tensor surface (laser_amd/tensor.py) against numpy semantics
and, for the chained GEMMs, the oracle.  Mirrors laser/tensor/{datatypes,initialization}.nim."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import __graft_entry__ as ge
    ge.build()
    import laser_amd
    return laser_amd


def test_new_tensor_metadata_and_zero_init(la):
    t = la.newTensor(np.float32, 3, 4, 5)
    assert (t.rank, t.size, t.shape, t.strides, t.offset) == (3, 60, (3, 4, 5), (20, 5, 1), 0)
    assert t.is_C_contiguous() and t.storage.memowner and t.unsafe_raw_data() % la.tensor.LASER_MEM_ALIGN == 0
    assert np.array_equal(t.to_numpy(), np.zeros((3, 4, 5), np.float32))
    assert la.newTensor(np.int64, (2, 0, 3)).to_numpy().shape == (2, 0, 3)
    with pytest.raises(ValueError):
        la.newTensor(np.float32, *([2] * 7))            # LASER_MAXRANK = 6
    with pytest.raises(TypeError):
        la.newTensor(np.float16, 4)


def test_to_tensor_and_views(la):
    ref = np.arange(2 * 3 * 4, dtype=np.int32).reshape(2, 3, 4)
    t = la.toTensor(ref.tolist(), dtype=np.int32)
    assert t.shape == (2, 3, 4) and np.array_equal(t.to_numpy(), ref)
    with pytest.raises(IndexError):
        la.toTensor([[1, 2], [3]])                      # ragged nesting (initialization.nim:184-189)
    v = t[1, ::2, 1:4]
    assert v.shape == (2, 3) and not v.is_C_contiguous()
    assert np.array_equal(v.to_numpy(), ref[1, ::2, 1:4])
    assert np.array_equal(t.transpose(2, 0, 1).to_numpy(), ref.transpose(2, 0, 1))
    assert np.array_equal(t[:, ::-1, :].to_numpy(), ref[:, ::-1, :])          # negative strides
    assert t[:, 1:2, :].is_C_contiguous() is False and t[0].is_C_contiguous()


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_deep_copy_copy_from_set_zero(la, dtype):
    rng = np.random.default_rng(3)
    ref = rng.integers(-1000, 1000, (6, 5, 7, 2, 3, 4)).astype(dtype)          # rank 6
    t = la.toTensor(ref)
    view = t.transpose(5, 0, 3, 1, 4, 2)[1:, ::2]
    c = la.deepCopy(view)
    assert c.is_C_contiguous() and c.storage is not t.storage
    assert np.array_equal(c.to_numpy(), ref.transpose(5, 0, 3, 1, 4, 2)[1:, ::2])
    # copyFrom into a view touches only what the view exposes
    big = la.toTensor(np.zeros((8, 9), dtype))
    src = la.toTensor(np.arange(12, dtype=dtype).reshape(3, 4))
    la.copyFrom(big[2:8:2, 1:9:2], src)
    want = np.zeros((8, 9), dtype); want[2:8:2, 1:9:2] = np.arange(12, dtype=dtype).reshape(3, 4)
    assert np.array_equal(big.to_numpy(), want)
    with pytest.raises(ValueError):
        la.copyFrom(big, src)
    la.copyFromRaw(big[0:2, 0:3], np.array([9, 8, 7, 6, 5, 4], dtype), 6)
    want[0:2, 0:3] = np.array([9, 8, 7, 6, 5, 4], dtype).reshape(2, 3)
    assert np.array_equal(big.to_numpy(), want)
    with pytest.raises(AssertionError):
        la.copyFromRaw(big, np.zeros(5, dtype), 5)
    with pytest.raises(ValueError):
        la.setZero(big[:, ::2])                          # "Input tensor is not contiguous."
    la.setZero(big[3])                                   # a contiguous row
    want[3] = 0
    assert np.array_equal(big.to_numpy(), want)
    d = la.newTensor(dtype, 1)
    old_storage = d.storage
    la.deepCopy(d, view)                                 # var-Tensor form: dst is re-bound, its old storage untouched
    assert d.shape == view.shape and d.storage is not old_storage and np.array_equal(d.to_numpy(), c.to_numpy())


def test_chained_gemm_stays_on_device_and_matches_oracle(la, oracle):
    rng = np.random.default_rng(11)
    M, K, N, P = 200, 600, 136, 72
    A = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
    Bt = rng.uniform(-0.1, 0.1, (N, K)).astype(np.float32)      # stored transposed
    E = rng.uniform(-0.1, 0.1, (M, P)).astype(np.float32)
    tA, tBt, tE = la.toTensor(A), la.toTensor(Bt), la.toTensor(E)
    C = la.matmul(tA, tBt.T)                                      # strided B view, device-resident result
    assert isinstance(C, la.Tensor) and C.shape == (M, N)
    D = la.matmul(C.T, tE)                                        # chained: consumes C through a transposed view
    isa = oracle.fused_isa(np.float32)
    C_ref = oracle.matmul(A, np.ascontiguousarray(Bt.T), isa=isa)
    D_ref = oracle.matmul(np.ascontiguousarray(C_ref.T), E, isa=isa)
    assert np.array_equal(C.to_numpy(), C_ref)
    assert np.array_equal(D.to_numpy(), D_ref)
    # physical transpose into a fresh tensor through the swapaxes primitive
    Ct = la.newTensor(np.float32, N, M)
    la.transpose2D_copy(Ct, C, M, N)
    assert np.array_equal(Ct.to_numpy(), C_ref.T)
    # gemm_strided straight on unsafe_raw_data-style operands with a bias + relu epilogue
    out = la.newTensor(np.float32, M, N)
    bias = la.toTensor(rng.uniform(-1, 1, (1, N)).astype(np.float32))
    la.gemm_strided(M, N, K, 1.0, tA, K, 1, tBt, 1, K, 0.0, out, N, 1, bias, 0, 1, "relu")
    assert np.array_equal(out.to_numpy(), oracle.apply_epilogue(C_ref, bias.to_numpy(), "relu"))


def test_conv_on_tensors_and_torch_interop(la, oracle):
    import torch
    rng = np.random.default_rng(5)
    ishape, kshape, pad, st = (2, 6, 14, 15), (8, 6, 3, 3), (1, 1), (1, 1)
    x = rng.uniform(0, 1, ishape).astype(np.float32); w = rng.uniform(0, 1, kshape).astype(np.float32)
    oshape = la.conv2d_out_shape(ishape, kshape, pad, st)
    out = la.newTensor(np.float32, *oshape)
    la.conv2d_im2col(out, oshape, la.toTensor(x), ishape, la.toTensor(w), kshape, pad, st, None)
    ref = oracle.conv2d_im2col(x, w, pad, st, isa=oracle.fused_isa(np.float32))
    assert np.array_equal(out.to_numpy(), ref)
    # zero-copy both ways
    tt = torch.as_tensor(out, device="cuda")
    assert tt.data_ptr() == out.unsafe_raw_data() and np.array_equal(tt.cpu().numpy(), ref)
    back = la.fromTorch(tt[:, 1:3])
    assert not back.storage.memowner and np.array_equal(back.to_numpy(), ref[:, 1:3])
    with pytest.raises(TypeError):
        la.matmul(la.toTensor(np.ones((2, 2), np.float32)), np.ones((2, 2), np.float32))   # mixed host / device


def test_storage_cache_reuses_and_rezeroes(la):
    """Freed storages are kept per size and handed out again -- zero-filled like a fresh allocShared0 block."""
    la.tensor.trimStorageCache()
    t = la.toTensor(np.full((513, 77), 7.0, np.float32))
    addr = t.unsafe_raw_data()
    del t
    u = la.newTensor(np.float32, 513, 77)                 # same size: the cached block comes back
    assert u.unsafe_raw_data() == addr
    assert np.array_equal(u.to_numpy(), np.zeros((513, 77), np.float32))
    v = la.newTensor(np.float32, 513, 78)                 # different size: a different block
    assert v.unsafe_raw_data() != addr
    del u, v
    la.tensor.trimStorageCache()
    w = la.newTensor(np.float64, 10)
    assert np.array_equal(w.to_numpy(), np.zeros(10))


@pytest.mark.gpu
def test_foreach_map_on_strided_rank6_views_vs_numpy():
    """The device twin of forEach (laser_hip_map_strided_*_dev): every op against numpy on rank-6 views with permuted,
    sliced and broadcast strides, in place and out of place, all four element types."""
    import laser_amd as la
    rng = np.random.default_rng(21)
    shape = (3, 4, 2, 5, 3, 6)
    for dtype in (np.float32, np.float64, np.int32, np.int64):
        if np.dtype(dtype).kind == "f":
            base_a = rng.uniform(0.1, 2.0, (4, 5, 3, 6, 3, 7)).astype(dtype)
            base_b = rng.uniform(0.1, 2.0, shape[::-1]).astype(dtype)
        else:
            base_a = rng.integers(-1000, 1000, (4, 5, 3, 6, 3, 7)).astype(dtype)
            base_b = rng.integers(-1000, 1000, shape[::-1]).astype(dtype)
        ta, tb = la.toTensor(base_a), la.toTensor(base_b)
        # a: a sliced + permuted view of a bigger tensor; b: a permuted view; dst: a strided slice of a bigger buffer
        va = ta[0:3, 0:4, 0:2, 0:5, 0:3, 0:6]
        na = base_a[0:3, 0:4, 0:2, 0:5, 0:3, 0:6]
        vb = tb.transpose()                 # all six axes reversed: strides ascending instead of descending
        nb = base_b.transpose()
        big = la.newTensor(dtype, 3, 4, 2, 5, 3, 12)
        vd = big[:, :, :, :, :, 0:12:2]
        fl = np.dtype(dtype).kind == "f"
        cases = [("copy", 1, lambda x, y: x), ("neg", 1, lambda x, y: -x), ("abs", 1, lambda x, y: np.abs(x)),
                 ("relu", 1, lambda x, y: np.maximum(x, 0)), ("square", 1, lambda x, y: x * x),
                 ("add", 2, lambda x, y: x + y), ("sub", 2, lambda x, y: x - y), ("mul", 2, lambda x, y: x * y),
                 ("max", 2, np.maximum), ("min", 2, np.minimum)]
        for op, nin, ref in cases:
            la.forEachMap(op, vd, va, vb if nin == 2 else None)
            got = big.to_numpy()[..., 0:12:2]
            want = ref(na, nb).astype(dtype)
            assert np.array_equal(got, want), (dtype, op)
            assert (big.to_numpy()[..., 1:12:2] == 0).all(), "the map wrote outside its strided destination"
        # parameters: scale / axpy / axpby / fill, with broadcast operands (a row vector against the rank-6 destination)
        row = la.toTensor(np.arange(6, dtype=dtype))
        la.forEachMap("scale", vd, va, alpha=3, beta=-2)
        assert np.array_equal(big.to_numpy()[..., 0:12:2], (3 * na - 2).astype(dtype))
        la.forEachMap("axpy", vd, row, vb, alpha=2)
        assert np.array_equal(big.to_numpy()[..., 0:12:2], (2 * np.arange(6, dtype=dtype) + nb).astype(dtype))
        la.forEachMap("axpby", vd, va, vb, alpha=2, beta=-1)
        assert np.array_equal(big.to_numpy()[..., 0:12:2], (2 * na - nb).astype(dtype))
        la.forEachMap("fill", vd, alpha=7)
        assert (big.to_numpy()[..., 0:12:2] == 7).all() and (big.to_numpy()[..., 1:12:2] == 0).all()
        # in place: dst aliases a
        tc = la.toTensor(np.ascontiguousarray(nb))
        la.forEachMap("add", tc, tc, vb)
        assert np.array_equal(tc.to_numpy(), nb + nb)
    # alpha / beta are of the element type: int64 values beyond 2^53 survive (they used to travel as doubles)
    t64 = la.newTensor(np.int64, 5)
    la.forEachMap("fill", t64, alpha=2 ** 60 + 1)
    assert (t64.to_numpy() == 2 ** 60 + 1).all()
    la.forEachMap("scale", t64, t64, alpha=1, beta=-(2 ** 60))
    assert (t64.to_numpy() == 1).all()
    # loud failures: an op code the library does not have; mismatched shapes do not broadcast
    ti = la.toTensor(np.arange(6, dtype=np.int32))
    with pytest.raises(la.LaserHipError):
        la.forEachMap(7, ti, ti)
    with pytest.raises(ValueError):
        la.forEachMap("add", ti, ti, la.toTensor(np.arange(5, dtype=np.int32)))
