"""Host-side mirror of Laser's primitives for the GEMM hot path, over the C-ABI of liblaser_hip.so.

Names, argument order and meaning follow the reference's Nim procs so that tests read like the
reference's own:

  gemm_strided            laser/primitives/matrix_multiplication/gemm.nim:184-193
  gemm_prepack{A,B}[_mem_required], gemm_packed     gemm_prepacked.nim:76-292
  transpose2D_copy, transpose2D_batched, nchw2nhwc, nhwc2nchw    laser/primitives/swapaxes.nim:16-112
  conv2d_out_shape, im2col_workspace_size, im2col, conv2d_im2col
                          benchmarks/convolution/conv2d_common.nim:15-45, conv2d_im2col.nim:10-166
  gemm (cblas-shaped)     benchmarks/third_party/blas.nim:12-23

A "raw buffer" argument is what the Nim side would pass as `ptr T` (the result of
`unsafe_raw_data`): either a numpy array (host memory -> blocking host-pointer entry point) or a
torch CUDA tensor (device memory -> `_dev` entry point on torch's current stream).  The pointer
passed down is the address of the array's first element; strides are in elements.  Nothing here
computes: every function forwards to the HIP library and raises LaserHipError on failure.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

try:  # torch is plumbing for device memory / streams only
    import torch
except Exception:  # pragma: no cover
    torch = None

_SFX = {"float32": "f32", "float64": "f64", "int32": "i32", "int64": "i64"}


def _is_lt(x):
    """A laser_amd.tensor.Tensor: Laser's Tensor[T] with device storage (laser_amd/tensor.py)."""
    return getattr(x, "_laser_hip_tensor", False)


def _is_dev(x):
    return _is_lt(x) or (torch is not None and isinstance(x, torch.Tensor))


def _sfx(x):
    name = str(x.dtype).replace("torch.", "")
    if name not in _SFX:
        raise TypeError(f"unsupported element type {x.dtype} (Laser GEMM: float32/float64/int32/int64)")
    return _SFX[name]


def _ptr(x):
    if x is None:
        return None
    if _is_lt(x):
        return C.c_void_p(x.unsafe_raw_data())   # what the Nim side passes: unsafe_raw_data(t)
    if _is_dev(x):
        if not x.is_cuda:
            raise TypeError("torch tensors must live on the GPU (pass numpy arrays for host memory)")
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


def _stream():
    if torch is None or not torch.cuda.is_available():
        return None  # the null stream
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _same_side(*xs):
    dev = [_is_dev(x) for x in xs if x is not None]
    if any(dev) and not all(dev):
        raise TypeError("operands must be all host (numpy) or all device (torch.cuda) buffers")
    return bool(dev) and dev[0]


def set_float_mode(mode):
    """0 = LASER_ORDER (bit-identical to Laser on an FMA host, default), 1 = FAST."""
    _lib.check(_lib.lib().laser_hip_set_float_mode(int(mode)))


def get_float_mode():
    return _lib.lib().laser_hip_get_float_mode()


_f32_config = -1   # what this process last asked for (the library has no getter; -1 = its heuristic)


def set_f32_config(cfg):
    """Force an fp32 tile configuration (index into f32_configs()); -1 = the library's heuristic."""
    global _f32_config
    _lib.check(_lib.lib().laser_hip_set_f32_config(int(cfg)))
    _f32_config = int(cfg)


def get_f32_config():
    """The configuration set_f32_config last selected in this process (-1 = heuristic)."""
    return _f32_config


def set_option(name, value):
    """laser_hip_set_option: every tuning / A-B switch of the library by name (include/laser_hip.h lists them)."""
    _lib.check(_lib.lib().laser_hip_set_option(name.encode(), int(value)))


def get_option(name):
    """laser_hip_get_option: an option's value, or a read-only diagnostic of the last launch ("last_f32_config", ...)."""
    v = C.c_int64()
    _lib.check(_lib.lib().laser_hip_get_option(name.encode(), C.byref(v)))
    return int(v.value)


# named conveniences over set_option / get_option (the tests and probes read better with them)
def set_f64_mfma(on):
    """True (default): float64 GEMM on the f64 matrix cores; False: VALU kernel."""
    set_option("f64_mfma", 1 if on else 0)


def set_f32_asm(mode):
    """1 (default): eligible float32 problems that fill the chip run the hand-scheduled assembly kernels; 0: never;
    2: whenever eligible, whatever the tile count (tests)."""
    set_option("f32_asm", int(mode))


def last_f32_asm():
    """0: the last float32 GEMM launch was a compiler-scheduled kernel; else 1 + the index of the assembly kernel
    (gemm_f32_asm.cpp: 1 / 2 laser-order / one-chain large tile, 3 / 4 the 128x128 tile, 5..8 the same with B transposed,
    9 / 10 one chain on the 256x128 tile)."""
    return get_option("last_f32_asm")


def plan_f32(M, N, K, laser_order=True, cus=256):
    """Diagnostics (no device needed): the kernel and launch plan the float32 launcher takes for a dense M x N x K product on a device
    of `cus` compute units (laser_hip_plan_f32): dict(kernel, plan, wgs, slices, tiles, tiles_m, tiles_n, slots); kernel 0 = the
    compiler-scheduled kernels take the problem; plan: "plain" / "cut" / "strided" / "hybrid" (strided whole rounds + a K-cut launch over
    the remaining tiles: `slices` are the second launch's)."""
    out = (C.c_int64 * 8)()
    _lib.check(_lib.lib().laser_hip_plan_f32(M, N, K, 1 if laser_order else 0, cus, out))
    return dict(kernel=int(out[0]), plan=("plain", "cut", "strided", "hybrid")[int(out[1])], wgs=int(out[2]), slices=int(out[3]), tiles=int(out[4]),
                tiles_m=int(out[5]), tiles_n=int(out[6]), slots=int(out[7]))


def set_i32_mfma(on):
    """True (default): int32 GEMM on the int8 matrix cores (limb decomposition); False: VALU kernel."""
    set_option("i32_mfma", 1 if on else 0)


def set_i64_mfma(on):
    """True (default): int64 GEMM on the int8 matrix cores (eight limbs, 36 products); False: VALU kernel."""
    set_option("i64_mfma", 1 if on else 0)


def set_conv_patch(on):
    """True (default): the implicit conv's B operand comes from an LDS-resident input patch where it fits."""
    set_option("conv_patch", 1 if on else 0)


def set_conv_kslice(on):
    """True (default): laser-order conv tail launches run Laser's kc slices in parallel + an ordered combine."""
    set_option("conv_kslice", 1 if on else 0)


def set_host_pipeline(mode):
    """bit 0 (default 1): large row-major host-pointer calls stream row panels x column panels; bit 1 set: the small
    zero-copy path synchronises its stream instead of polling completion flags."""
    set_option("host_pipeline_2d", int(mode) & 1)
    set_option("zero_copy_poll", 0 if int(mode) & 2 else 1)


def set_slice_parallel(on):
    """True (default): few-tile / long-K float problems run Laser's kc slices in parallel + an ordered combine.
    (tuning: 2..100 sets the minimum slice count, > 100 the tile-count threshold)"""
    on = int(on)
    set_option("slice_parallel", 1 if on else 0)
    if on > 100:
        set_option("slice_parallel_tiles", on)
    elif on >= 2:
        set_option("slice_parallel_min", on)


def set_split_tail(on):
    """True (default): fp32 problems whose last round of tiles would be badly filled run as main + tail launches."""
    set_option("split_tail", 1 if on else 0)


def last_split():
    """Column where the last float GEMM / conv launch was cut into main + tail (0: one launch)."""
    return get_option("last_split")


def set_small_path(on):
    """True (default): small / batched-tiny float problems run the one-wave-per-block small-matrix kernel."""
    set_option("small_path", 1 if on else 0)


def set_skinny(on):
    """True (default): M <= 8 or N <= 8 float problems run the streaming (matrix-vector) kernel."""
    set_option("skinny", 1 if on else 0)


def last_f32_config():
    """Index into f32_configs() of the tile configuration the last fp32 GEMM / conv launch used."""
    return get_option("last_f32_config")


def set_conv_implicit(on):
    """True (default): implicit-GEMM convolution; False: explicit im2col workspace + batched GEMM."""
    set_option("conv_implicit", 1 if on else 0)


def f32_configs():
    L = _lib.lib()
    return [L.laser_hip_f32_config_name(i).decode() for i in range(L.laser_hip_f32_config_count())]


# ---- GEMM ----------------------------------------------------------------------------------------
ACTIVATIONS = {None: 0, "none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}


def _activation_code(activation):
    if isinstance(activation, int):
        return activation
    try:
        return ACTIVATIONS[activation]
    except KeyError:
        raise ValueError(f"unknown activation {activation!r} (one of {sorted(k for k in ACTIVATIONS if k)})")


PRE_RELU_A, PRE_RELU_B = 0x100, 0x200      # include/laser_hip.h LASER_HIP_PRE_RELU_A / _B: ORed into the activation code


def gemm_strided(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C_,
                 rowStrideC, colStrideC, bias=None, rowStrideBias=0, colStrideBias=0, activation=None, pre=0):
    """C <- alpha*A*B + beta*C, element X[r,c] at X_ptr[r*rowStride + c*colStride].

    Fused epilogue (what the reference plans, README.md:238-242): with `bias` (a strided M x N view whose
    strides may be 0) and/or `activation` ("relu" | "tanh" | "sigmoid"),
    C <- act(alpha*A*B + beta*C + bias), applied once on the accumulator before the store; float32/float64.
    Fused prologue (README.md:243-244): pre = PRE_RELU_A and / or PRE_RELU_B takes the product of relu(A) / relu(B)."""
    L = _lib.lib()
    s = _sfx(C_)
    if _sfx(A) != s or _sfx(B) != s:
        raise TypeError("A, B, C must share one element type")
    ct = _lib.ctype_of(s)
    args = [M, N, K, ct(alpha), _ptr(A), rowStrideA, colStrideA, _ptr(B), rowStrideB, colStrideB,
            ct(beta), _ptr(C_), rowStrideC, colStrideC]
    act = _activation_code(activation) | int(pre)
    if bias is not None or act:
        if s not in ("f32", "f64"):
            raise TypeError("the fused epilogue is float32/float64 only")
        if bias is not None and _sfx(bias) != s:
            raise TypeError("bias must have the element type of C")
        args += [_ptr(bias), rowStrideBias, colStrideBias, act]
        if _same_side(A, B, C_, bias):
            _lib.check(getattr(L, f"laser_hip_gemm_strided_ex_{s}_dev")(*args, _stream()))
        else:
            _lib.check(getattr(L, f"laser_hip_gemm_strided_ex_{s}")(*args))
        return C_
    if _same_side(A, B, C_):
        _lib.check(getattr(L, f"laser_hip_gemm_strided_{s}_dev")(*args, _stream()))
    else:
        _lib.check(getattr(L, f"laser_hip_gemm_strided_{s}")(*args))
    return C_


def gemm_strided_batched(batch, M, N, K, alpha, A, rsA, csA, bsA, B, rsB, csB, bsB, beta, C_, rsC, csC, bsC):
    """Device-only: `batch` independent problems, operand b at ptr + b*batchStride (0 = shared)."""
    L = _lib.lib()
    s = _sfx(C_)
    if not _same_side(A, B, C_):
        raise TypeError("batched GEMM is a device-resident entry point")
    ct = _lib.ctype_of(s)
    _lib.check(getattr(L, f"laser_hip_gemm_strided_batched_{s}_dev")(
        batch, M, N, K, ct(alpha), _ptr(A), rsA, csA, bsA, _ptr(B), rsB, csB, bsB, ct(beta), _ptr(C_),
        rsC, csC, bsC, _stream()))
    return C_


def _estrides(x):
    if _is_lt(x):
        return tuple(x.strides)
    if _is_dev(x):
        return tuple(x.stride())
    return tuple(st // x.dtype.itemsize for st in x.strides)


def matmul(A, B, alpha=1, beta=0, out=None, bias=None, activation=None, pre=0):
    """Convenience over gemm_strided for 2-D views of any strides (numpy or torch.cuda).
    `bias`: a vector of N values (one per column, a dense layer's bias) or an (M, N) / broadcastable 2-D view."""
    M, K = A.shape
    K2, N = B.shape
    if K != K2:
        raise ValueError("inner dimensions differ")
    if out is None:
        if _is_lt(A):
            from .tensor import newTensor
            out = newTensor(A.dtype, M, N)
        else:
            out = torch.zeros((M, N), dtype=A.dtype, device=A.device) if _is_dev(A) else np.zeros((M, N), dtype=A.dtype)
    (rsA, csA), (rsB, csB), (rsC, csC) = _estrides(A), _estrides(B), _estrides(out)
    rb = cb = 0
    if bias is not None:
        if len(bias.shape) == 1:
            if bias.shape[0] != N:
                raise ValueError("a 1-D bias must have N entries")
            cb = _estrides(bias)[0]
        else:
            (r_, c_) = bias.shape
            if r_ not in (1, M) or c_ not in (1, N):
                raise ValueError("bias does not broadcast to (M, N)")
            rb, cb = _estrides(bias)
            rb, cb = (rb if r_ == M and M > 1 else 0), (cb if c_ == N and N > 1 else 0)
    gemm_strided(M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, out, rsC, csC, bias, rb, cb, activation, pre)
    return out


def _np_dtype(T):
    return np.dtype(T) if not (torch is not None and isinstance(T, torch.dtype)) else np.dtype(str(T).replace("torch.", ""))


def gemm_prepackA_mem_required(T, M, N, K):
    return getattr(_lib.lib(), f"laser_hip_gemm_prepackA_mem_required_{_SFX[_np_dtype(T).name]}")(M, N, K)


def gemm_prepackB_mem_required(T, M, N, K):
    return getattr(_lib.lib(), f"laser_hip_gemm_prepackB_mem_required_{_SFX[_np_dtype(T).name]}")(M, N, K)


def _prepack(which, dst, M, N, K, src, rs, cs):
    L = _lib.lib()
    s = _sfx(src)
    if _same_side(dst, src):
        _lib.check(getattr(L, f"laser_hip_gemm_prepack{which}_{s}_dev")(_ptr(dst), M, N, K, _ptr(src), rs, cs, _stream()))
    else:
        _lib.check(getattr(L, f"laser_hip_gemm_prepack{which}_{s}")(_ptr(dst), M, N, K, _ptr(src), rs, cs))
    return dst


def gemm_prepackA(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA):
    return _prepack("A", dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)


def gemm_prepackB(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB):
    return _prepack("B", dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)


def gemm_packed(M, N, K, alpha, packedA, packedB, beta, C_, rowStrideC, colStrideC):
    L = _lib.lib()
    s = _sfx(C_)
    ct = _lib.ctype_of(s)
    if _same_side(packedA, packedB, C_):
        _lib.check(getattr(L, f"laser_hip_gemm_packed_{s}_dev")(M, N, K, ct(alpha), _ptr(packedA), _ptr(packedB),
                                                                ct(beta), _ptr(C_), rowStrideC, colStrideC, _stream()))
    else:
        _lib.check(getattr(L, f"laser_hip_gemm_packed_{s}")(M, N, K, ct(alpha), _ptr(packedA), _ptr(packedB),
                                                            ct(beta), _ptr(C_), rowStrideC, colStrideC))
    return C_


def gemm_prepack_release(packed):
    _lib.check(_lib.lib().laser_hip_gemm_prepack_release(_ptr(packed)))


def aligned_host_buffer(nbytes, dtype=np.uint8, align=64):
    """A 64-byte aligned numpy buffer (what `newTensor` / Laser's allocator gives the Nim caller)."""
    raw = np.zeros(nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes].view(dtype)


def pinned_host_buffer(shape, dtype=np.float32):
    """A numpy array over pinned host memory from laser_hip_host_alloc (what a tensor allocator would use to give the
    host-pointer entry points direct-DMA uploads); freed when the array is garbage-collected."""
    import weakref
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _lib.check(_lib.lib().laser_hip_host_alloc(C.byref(p), max(n, 1)))
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, _lib.lib().laser_hip_host_free, C.c_void_p(p.value))
    return arr


def host_register(arr):
    """Pin an existing numpy array's memory (laser_hip_host_register); pair with host_unregister."""
    _lib.check(_lib.lib().laser_hip_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes))


def host_unregister(arr):
    _lib.check(_lib.lib().laser_hip_host_unregister(C.c_void_p(arr.ctypes.data)))


# ---- transposes ------------------------------------------------------------------------------------
def _bsfx(x):
    size = x.element_size() if (_is_dev(x) and not _is_lt(x)) else x.dtype.itemsize
    if size not in (1, 2, 4, 8):
        raise TypeError("transposes support 1-, 2-, 4- and 8-byte elements")
    return {4: "b32", 8: "b64", 2: "b16", 1: "b8"}[size]


def transpose2D_batched(dst, src, N, NR, NC):
    L = _lib.lib()
    b = _bsfx(src)
    if _same_side(dst, src):
        _lib.check(getattr(L, f"laser_hip_transpose2d_batched_{b}_dev")(_ptr(dst), _ptr(src), N, NR, NC, _stream()))
    else:
        _lib.check(getattr(L, f"laser_hip_transpose2d_batched_{b}")(_ptr(dst), _ptr(src), N, NR, NC))
    return dst


def transpose2D_copy(dst, src, NR, NC):
    if _same_side(dst, src):
        return transpose2D_batched(dst, src, 1, NR, NC)
    _lib.check(getattr(_lib.lib(), f"laser_hip_transpose2d_copy_{_bsfx(src)}")(_ptr(dst), _ptr(src), NR, NC))
    return dst


def nchw2nhwc(dst_nhwc, src_nchw, N, C_, H, W):
    if _same_side(dst_nhwc, src_nchw):
        return transpose2D_batched(dst_nhwc, src_nchw, N, C_, H * W)
    _lib.check(getattr(_lib.lib(), f"laser_hip_nchw2nhwc_{_bsfx(src_nchw)}")(_ptr(dst_nhwc), _ptr(src_nchw), N, C_, H, W))
    return dst_nhwc


def nhwc2nchw(dst_nchw, src_nhwc, N, C_, H, W):
    if _same_side(dst_nchw, src_nhwc):
        return transpose2D_batched(dst_nchw, src_nhwc, N, H * W, C_)
    _lib.check(getattr(_lib.lib(), f"laser_hip_nhwc2nchw_{_bsfx(src_nhwc)}")(_ptr(dst_nchw), _ptr(src_nhwc), N, C_, H, W))
    return dst_nchw


# ---- convolution -------------------------------------------------------------------------------------
# TensorShape = (n, c, h, w); KernelShape = (c_out, c_in, kH, kW); Padding = (h, w); Strides = (h, w)
def conv2d_out_shape(ishape, kshape, padding, strides):
    o = [C.c_int64() for _ in range(4)]
    _lib.check(_lib.lib().laser_hip_conv2d_out_shape(*ishape, *kshape, *padding, *strides, *[C.byref(v) for v in o]))
    return tuple(v.value for v in o)


def im2col_workspace_size(ishape, kshape, padding, strides):
    return _lib.lib().laser_hip_im2col_workspace_size(*ishape, *kshape, *padding, *strides)


def _f32_dense(name, x, min_elems, dtype="float32"):
    """The conv / im2col / cblas mirrors pass raw pointers of dense float32 buffers: refuse anything else instead of
    reinterpreting it (a float64 or non-contiguous array would silently be read as garbage, or out of bounds)."""
    if x is None:
        return
    dt = str(x.dtype).replace("torch.", "")
    if dt != dtype:
        raise TypeError(f"{name}: {dtype} expected, got {x.dtype}")
    contiguous = x.is_contiguous() if (_is_dev(x) and not _is_lt(x)) else (x.is_C_contiguous() if _is_lt(x) else x.flags["C_CONTIGUOUS"])
    if not contiguous:
        raise ValueError(f"{name}: a dense (C-contiguous) buffer is required")
    n = x.numel() if (_is_dev(x) and not _is_lt(x)) else int(x.size)
    if n < min_elems:
        raise ValueError(f"{name}: holds {n} elements, {min_elems} are needed")


def im2col(pworkspace, oshape, pinput, ishape, kshape, padding, strides):
    """One image [c,h,w] -> pworkspace [c*kH*kW, oH*oW]; float32 or float64 (im2col*[T], conv2d_im2col.nim:42-50)."""
    L = _lib.lib()
    dt = str(pinput.dtype).replace("torch.", "")
    if dt not in ("float32", "float64"):
        raise TypeError(f"pinput: float32 or float64 expected, got {pinput.dtype}")
    sfx = "f32" if dt == "float32" else "f64"
    _f32_dense("pworkspace", pworkspace, ishape[1] * kshape[2] * kshape[3] * oshape[2] * oshape[3], dt)
    _f32_dense("pinput", pinput, ishape[1] * ishape[2] * ishape[3], dt)
    if _same_side(pworkspace, pinput):
        _lib.check(getattr(L, f"laser_hip_im2col_{sfx}_dev")(_ptr(pworkspace), oshape[2], oshape[3], _ptr(pinput), 1, ishape[1],
                                                             ishape[2], ishape[3], kshape[2], kshape[3], *padding, *strides, _stream()))
    else:
        _lib.check(getattr(L, f"laser_hip_im2col_{sfx}")(_ptr(pworkspace), oshape[2], oshape[3], _ptr(pinput), ishape[1], ishape[2],
                                                         ishape[3], kshape[2], kshape[3], *padding, *strides))
    return pworkspace


def conv2d_im2col(output, oshape, input_, ishape, kernel, kshape, padding, strides, pworkspace=None,
                  bias=None, activation=None):
    """conv2d_im2col (benchmarks/convolution/conv2d_im2col.nim:90-166); `bias` ([c_out]) and `activation`
    are the fused epilogue: output <- act(conv + bias[c_out]) in the same kernel."""
    L = _lib.lib()
    if tuple(oshape) != conv2d_out_shape(ishape, kshape, padding, strides):
        raise ValueError("oshape does not match conv2d_out_shape(ishape, kshape, padding, strides)")
    if oshape[1] != kshape[0]:
        raise ValueError("oshape.c != kshape.c_out")  # conv2d_im2col.nim:109
    _f32_dense("output", output, math.prod(oshape))
    _f32_dense("input", input_, math.prod(ishape))
    _f32_dense("kernel", kernel, math.prod(kshape))
    _f32_dense("pworkspace", pworkspace, im2col_workspace_size(ishape, kshape, padding, strides))   # ONE image's worth (the reference's contract)
    _f32_dense("bias", bias, kshape[0])
    args = [_ptr(output), _ptr(input_), *ishape, _ptr(kernel), *kshape, *padding, *strides, _ptr(pworkspace)]
    act = _activation_code(activation)
    if bias is not None or act:
        if bias is not None and math.prod(bias.shape) != kshape[0]:
            raise ValueError("bias must hold c_out values")
        args += [_ptr(bias), act]
        if _same_side(output, input_, kernel, pworkspace, bias):
            _lib.check(L.laser_hip_conv2d_im2col_ex_f32_dev(*args, _stream()))
        else:
            _lib.check(L.laser_hip_conv2d_im2col_ex_f32(*args))
        return output
    if _same_side(output, input_, kernel, pworkspace):
        _lib.check(L.laser_hip_conv2d_im2col_f32_dev(*args, _stream()))
    else:
        _lib.check(L.laser_hip_conv2d_im2col_f32(*args))
    return output


# cblas enums (benchmarks/third_party/blas.nim:12-16)
noTranspose, transpose, conjTranspose = 111, 112, 113
rowMajor, colMajor = 101, 102


def gemm(ORDER, TRANSA, TRANSB, M, N, K, ALPHA, A, LDA, B, LDB, BETA, C_, LDC):
    """cblas-shaped gemm, the call conv2d_im2col makes in the reference (conv2d_im2col.nim:161-166)."""
    L = _lib.lib()
    s = _sfx(C_)
    if s not in ("f32", "f64") or _sfx(A) != s or _sfx(B) != s:
        raise TypeError("cblas-shaped gemm: A, B, C must all be float32 or all float64 (blas.nim:18-23)")
    if _is_dev(A) or _is_dev(B) or _is_dev(C_):
        raise TypeError("cblas-shaped gemm is a host-pointer entry point (numpy buffers)")
    rowm = ORDER == rowMajor
    for name, x, ld, r, c in (("A", A, LDA, (M if TRANSA == noTranspose else K), (K if TRANSA == noTranspose else M)),
                              ("B", B, LDB, (K if TRANSB == noTranspose else N), (N if TRANSB == noTranspose else K)),
                              ("C", C_, LDC, M, N)):
        outer, inner = (r, c) if rowm else (c, r)
        if ld < inner:
            raise ValueError(f"{name}: leading dimension {ld} < {inner}")
        if int(x.size) < (outer - 1) * ld + inner:
            raise ValueError(f"{name}: buffer too small for {outer} x {inner} with leading dimension {ld}")
    fn = {"f32": L.laser_hip_cblas_sgemm, "f64": L.laser_hip_cblas_dgemm}[s]
    _lib.check(fn(ORDER, TRANSA, TRANSB, M, N, K, ALPHA, _ptr(A), LDA, _ptr(B), LDB, BETA, _ptr(C_), LDC))
    return C_
