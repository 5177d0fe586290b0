"""ctypes/numpy front-end for the CPU oracle (oracle/liblaser_oracle.so).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/laser_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from laser_amd/.

The function names and argument order follow the reference's Nim procs
(laser/primitives/matrix_multiplication/gemm.nim:184-193, gemm_prepacked.nim:76-292,
laser/primitives/swapaxes.nim:16-112, benchmarks/convolution/conv2d_im2col.nim:10-100).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblaser_oracle.so")

ISA_GENERIC, ISA_SSE, ISA_SSE2, ISA_SSE4_1, ISA_AVX, ISA_AVX_FMA, ISA_AVX2, ISA_AVX512 = range(8)

_DT = {
    np.dtype(np.float32): ("f32", C.c_float, 0),
    np.dtype(np.float64): ("f64", C.c_double, 1),
    np.dtype(np.int32): ("i32", C.c_int32, 2),
    np.dtype(np.int64): ("i64", C.c_int64, 3),
}


def build(force=False):
    """Compile the oracle with oracle/Makefile (gcc).  Building the checker is not using it."""
    src_newer = (not os.path.exists(_SO)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("laser_oracle.c", "laser_gemm_impl.inc", "Makefile"))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        i64, vp, ci = C.c_int64, C.c_void_p, C.c_int
        for sfx, ct, _ in _DT.values():
            getattr(_lib, f"oracle_gemm_strided_{sfx}").argtypes = [
                i64, i64, i64, ct, vp, i64, i64, vp, i64, i64, ct, vp, i64, i64, ci, ci]
            for ab in "AB":
                f = getattr(_lib, f"oracle_gemm_prepack{ab}_mem_required_{sfx}")
                f.argtypes = [i64, i64, i64, ci]
                f.restype = i64
                getattr(_lib, f"oracle_gemm_prepack{ab}_{sfx}").argtypes = [vp, i64, i64, i64, vp, i64, i64, ci]
            getattr(_lib, f"oracle_gemm_packed_{sfx}").argtypes = [i64, i64, i64, ct, vp, vp, ct, vp, i64, i64, ci]
            getattr(_lib, f"oracle_transpose2D_copy_{sfx}").argtypes = [vp, vp, i64, i64]
            getattr(_lib, f"oracle_transpose2D_batched_{sfx}").argtypes = [vp, vp, i64, i64, i64]
            getattr(_lib, f"oracle_nchw2nhwc_{sfx}").argtypes = [vp, vp, i64, i64, i64, i64]
            getattr(_lib, f"oracle_nhwc2nchw_{sfx}").argtypes = [vp, vp, i64, i64, i64, i64]
        _lib.oracle_im2col_workspace_size.argtypes = [i64] * 9
        _lib.oracle_im2col_workspace_size.restype = i64
        _lib.oracle_im2col_f32.argtypes = [vp, i64, i64, vp] + [i64] * 9
        _lib.oracle_conv2d_im2col_f32.argtypes = [vp, vp, i64, i64, i64, i64, vp] + [i64] * 7 + [vp, ci]
        _lib.oracle_conv2d_direct_f32.argtypes = [vp, vp, i64, i64, i64, i64, vp] + [i64] * 7
        _lib.oracle_mean_relative_error_f32.argtypes = [vp, vp, i64]
        _lib.oracle_mean_relative_error_f32.restype = C.c_float
        _lib.oracle_naive_gemm_f64acc.argtypes = [i64, i64, i64, vp, i64, i64, vp, i64, i64, vp]
        _lib.oracle_detect_isa.argtypes = [ci]
        _lib.oracle_set_num_threads.argtypes = [ci]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def detect_isa(dtype):
    return lib().oracle_detect_isa(_DT[np.dtype(dtype)][2])


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def fused_isa(dtype):
    """The ISA a modern host would pick; for floats always an FMA one (AVX512 or AVX_FMA), which is
    what the MI355X f32 MFMA (a k-ordered fmaf chain) is bit-comparable with."""
    isa = detect_isa(dtype)
    if np.dtype(dtype).kind == "f" and isa not in (ISA_AVX512, ISA_AVX_FMA):
        isa = ISA_AVX_FMA
    return isa


def gemm_strided(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C_,
                 rowStrideC, colStrideC, isa=None, use_simd=True):
    """C <- alpha*A*B + beta*C on raw buffers with element strides (gemm.nim:184-193).
    A, B, C_ are numpy arrays used as flat buffers (element [r, c] at r*rowStride + c*colStride)."""
    dt = np.dtype(C_.dtype)
    sfx, ct, _ = _DT[dt]
    assert A.dtype == dt and B.dtype == dt
    if isa is None:
        isa = fused_isa(dt)
    rc = getattr(lib(), f"oracle_gemm_strided_{sfx}")(
        M, N, K, ct(alpha), _ptr(A), rowStrideA, colStrideA, _ptr(B), rowStrideB, colStrideB,
        ct(beta), _ptr(C_), rowStrideC, colStrideC, isa, int(use_simd))
    if rc:
        raise RuntimeError(f"oracle_gemm_strided_{sfx} -> {rc}")
    return C_


def matmul(A, B, alpha=1, beta=0, C_=None, isa=None, use_simd=True):
    """Convenience: 2-D numpy views (any strides) -> dense result via the oracle."""
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    dt = A.dtype
    if C_ is None:
        C_ = np.zeros((M, N), dtype=dt)
    it = dt.itemsize

    def base(x):
        return x  # numpy passes the address of element [0,0]

    gemm_strided(M, N, K, alpha, base(A), A.strides[0] // it, A.strides[1] // it, base(B),
                 B.strides[0] // it, B.strides[1] // it, beta, C_, C_.strides[0] // it,
                 C_.strides[1] // it, isa=isa, use_simd=use_simd)
    return C_


def _aligned(nbytes, dtype):
    raw = np.zeros(nbytes + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 64
    return raw[off:off + nbytes].view(dtype)


def gemm_prepackB_mem_required(dtype, M, N, K, isa=None):
    sfx = _DT[np.dtype(dtype)][0]
    return getattr(lib(), f"oracle_gemm_prepackB_mem_required_{sfx}")(M, N, K, fused_isa(dtype) if isa is None else isa)


def gemm_prepackA_mem_required(dtype, M, N, K, isa=None):
    sfx = _DT[np.dtype(dtype)][0]
    return getattr(lib(), f"oracle_gemm_prepackA_mem_required_{sfx}")(M, N, K, fused_isa(dtype) if isa is None else isa)


def gemm_prepack_and_run(A, B, alpha=1, beta=0, C_=None, isa=None):
    """prepackA + prepackB + gemm_packed, the reference's pack_and_test (gemm_prepacked.nim:314-351)."""
    dt = A.dtype
    sfx, ct, _ = _DT[np.dtype(dt)]
    M, K = A.shape
    N = B.shape[1]
    isa = fused_isa(dt) if isa is None else isa
    pa = _aligned(gemm_prepackA_mem_required(dt, M, N, K, isa), dt)
    pb = _aligned(gemm_prepackB_mem_required(dt, M, N, K, isa), dt)
    it = dt.itemsize
    L = lib()
    assert getattr(L, f"oracle_gemm_prepackA_{sfx}")(_ptr(pa), M, N, K, _ptr(A), A.strides[0] // it, A.strides[1] // it, isa) == 0
    assert getattr(L, f"oracle_gemm_prepackB_{sfx}")(_ptr(pb), M, N, K, _ptr(B), B.strides[0] // it, B.strides[1] // it, isa) == 0
    if C_ is None:
        C_ = np.zeros((M, N), dtype=dt)
    getattr(L, f"oracle_gemm_packed_{sfx}")(M, N, K, ct(alpha), _ptr(pa), _ptr(pb), ct(beta), _ptr(C_),
                                            C_.strides[0] // it, C_.strides[1] // it, isa)
    return C_


def transpose2D_copy(src):
    src = np.ascontiguousarray(src)
    NR, NC = src.shape
    dst = np.empty((NC, NR), dtype=src.dtype)
    getattr(lib(), f"oracle_transpose2D_copy_{_DT[src.dtype][0]}")(_ptr(dst), _ptr(src), NR, NC)
    return dst


def transpose2D_batched(src):
    src = np.ascontiguousarray(src)
    N, NR, NC = src.shape
    dst = np.empty((N, NC, NR), dtype=src.dtype)
    getattr(lib(), f"oracle_transpose2D_batched_{_DT[src.dtype][0]}")(_ptr(dst), _ptr(src), N, NR, NC)
    return dst


def nchw2nhwc(src):
    src = np.ascontiguousarray(src)
    N, Cc, H, W = src.shape
    dst = np.empty((N, H, W, Cc), dtype=src.dtype)
    getattr(lib(), f"oracle_nchw2nhwc_{_DT[src.dtype][0]}")(_ptr(dst), _ptr(src), N, Cc, H, W)
    return dst


def nhwc2nchw(src):
    src = np.ascontiguousarray(src)
    N, H, W, Cc = src.shape
    dst = np.empty((N, Cc, H, W), dtype=src.dtype)
    getattr(lib(), f"oracle_nhwc2nchw_{_DT[src.dtype][0]}")(_ptr(dst), _ptr(src), N, Cc, H, W)
    return dst


def conv2d_out_shape(ishape, kshape, padding, strides):
    n, c, h, w = ishape
    co, ci, kh, kw = kshape
    return (n, co, 1 + (h + 2 * padding[0] - kh) // strides[0], 1 + (w + 2 * padding[1] - kw) // strides[1])


def im2col_workspace_size(ishape, kshape, padding, strides):
    n, c, h, w = ishape
    return lib().oracle_im2col_workspace_size(c, h, w, kshape[2], kshape[3], padding[0], padding[1], strides[0], strides[1])


def im2col(image, kshape, padding, strides):
    """One NCHW image [C,H,W] -> [C*kH*kW, outH*outW] (conv2d_im2col.nim:42-88)."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    c, h, w = image.shape
    _, _, oh, ow = conv2d_out_shape((1, c, h, w), kshape, padding, strides)
    ws = np.empty((c * kshape[2] * kshape[3], oh * ow), dtype=np.float32)
    lib().oracle_im2col_f32(_ptr(ws), oh, ow, _ptr(image), c, h, w, kshape[2], kshape[3],
                            padding[0], padding[1], strides[0], strides[1])
    return ws


def conv2d_im2col(input_, kernel, padding, strides, isa=None):
    input_ = np.ascontiguousarray(input_, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    oshape = conv2d_out_shape(input_.shape, kernel.shape, padding, strides)
    out = np.zeros(oshape, dtype=np.float32)
    ws = np.empty(max(1, im2col_workspace_size(input_.shape, kernel.shape, padding, strides)), dtype=np.float32)
    n, c, h, w = input_.shape
    co, ci, kh, kw = kernel.shape
    rc = lib().oracle_conv2d_im2col_f32(_ptr(out), _ptr(input_), n, c, h, w, _ptr(kernel), co, kh, kw,
                                        padding[0], padding[1], strides[0], strides[1], _ptr(ws),
                                        fused_isa(np.float32) if isa is None else isa)
    assert rc == 0
    return out


def conv2d_direct(input_, kernel, padding, strides):
    input_ = np.ascontiguousarray(input_, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    oshape = conv2d_out_shape(input_.shape, kernel.shape, padding, strides)
    out = np.zeros(oshape, dtype=np.float32)
    n, c, h, w = input_.shape
    co, ci, kh, kw = kernel.shape
    lib().oracle_conv2d_direct_f32(_ptr(out), _ptr(input_), n, c, h, w, _ptr(kernel), co, kh, kw,
                                   padding[0], padding[1], strides[0], strides[1])
    return out


def apply_epilogue(C_, bias=None, activation=None):
    """The fused epilogue the reference plans (README.md:238-242; TODO gemm_ukernel_generic.nim:78-79),
    restated on top of a finished gemm result: one more rounding for `+ bias` (numpy broadcasting),
    then the activation, all in C's own precision.  relu = max(x, 0) with NaN -> 0 like `x > 0 ? x : 0`."""
    out = C_.copy()
    if bias is not None:
        out = (out + np.asarray(bias, dtype=out.dtype)).astype(out.dtype)
    one = out.dtype.type(1)
    if activation in (None, "none", 0):
        pass
    elif activation in ("relu", 1):
        out = np.where(out > 0, out, out.dtype.type(0)).astype(out.dtype)
    elif activation in ("tanh", 2):
        out = np.tanh(out).astype(out.dtype)
    elif activation in ("sigmoid", 3):
        out = (one / (one + np.exp(-out))).astype(out.dtype)
    else:
        raise ValueError(activation)
    return out


def mean_relative_error(y, y_true):
    """laser/private/error_functions.nim:6-26, float32 accumulation like the reference."""
    y = np.ascontiguousarray(y, dtype=np.float32).ravel()
    y_true = np.ascontiguousarray(y_true, dtype=np.float32).ravel()
    return float(lib().oracle_mean_relative_error_f32(_ptr(y), _ptr(y_true), y.size))


def naive_gemm_f64(A, B):
    A = np.asarray(A, dtype=np.float32)
    B = np.asarray(B, dtype=np.float32)
    M, K = A.shape
    N = B.shape[1]
    out = np.empty((M, N), dtype=np.float64)
    lib().oracle_naive_gemm_f64acc(M, N, K, _ptr(A), A.strides[0] // 4, A.strides[1] // 4, _ptr(B),
                                   B.strides[0] // 4, B.strides[1] // 4, _ptr(out))
    return out
