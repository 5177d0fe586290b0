"""Device-resident twin of Laser's Tensor[T] (laser/tensor/datatypes.nim:18-30) -- SURVEY.md section 8f rank 3.

    Tensor[T] = object            shape, strides: Metadata (elements, rank <= LASER_MAXRANK = 6)
                                  offset: int
                                  storage: CpuStorage[T]      raw_buffer / memalloc / memowner

Here `storage` is a HipStorage whose raw_buffer is a DEVICE address (HBM), so chains of
gemm_strided / transposes / conv on Tensors never cross PCIe; everything else keeps the reference's
names and behaviour:

    rank, size, is_C_contiguous, unsafe_raw_data            datatypes.nim:32-83
    newTensor (zero-initialised), toTensor                  initialization.nim:156-202
    deepCopy, copyFrom, copyFromRaw, setZero                initialization.nim:42-154

The procs that take `var Tensor` in Nim mutate their first argument here as well.  Slicing
(`t[1:3, ::2]`) and `transpose()` return views on the same storage, which is how the strided entry
points get exercised.  Nothing in this module computes: data movement goes through liblaser_hip.so
(laser_hip_storage_*, laser_hip_copy_strided_*_dev), and the module fails loudly without a GPU.
"""
import ctypes as C

import numpy as np

from . import _lib

LASER_MAXRANK = 6        # laser/dynamic_stack_arrays.nim:6
LASER_MEM_ALIGN = 64     # laser/compiler_optim_hints.nim:6
_DTYPES = ("float32", "float64", "int32", "int64")


def _stream():
    try:
        import torch
        if torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    except Exception:  # pragma: no cover
        pass
    return None


class HipStorage:
    """CpuStorage's twin (datatypes.nim:24-30, allocator.nim:11-29): `raw_buffer` is a device address,
    `memowner` says whether the finalizer frees it."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        # zero fill ordered on the stream the tensor's kernels will run on (torch side streams are non-blocking)
        _lib.check(_lib.lib().laser_hip_storage_alloc_stream(C.byref(p), int(nbytes), _stream()))
        self.raw_buffer = p.value or 0
        self.memalloc = self.raw_buffer
        self.memowner = True
        self.nbytes = int(nbytes)
        self._keepalive = None

    @classmethod
    def external(cls, address, nbytes, keepalive=None):
        """Non-owning storage over memory someone else allocated (e.g. a torch CUDA tensor)."""
        s = cls.__new__(cls)
        s.raw_buffer = int(address)
        s.memalloc = 0
        s.memowner = False
        s.nbytes = int(nbytes)
        s._keepalive = keepalive
        return s

    def __del__(self):  # the storage finalizer
        if getattr(self, "memowner", False) and self.memalloc:
            try:
                _lib.lib().laser_hip_storage_free(C.c_void_p(self.memalloc))
            except Exception:  # pragma: no cover  (interpreter shutdown)
                pass
            self.memalloc = 0


def _row_major_strides(shape):
    strides, size = [0] * len(shape), 1
    for i in range(len(shape) - 1, -1, -1):  # initTensorMetadataImpl, initialization.nim:24-32
        strides[i] = size
        size *= shape[i]
    return tuple(strides), size


class Tensor:
    _laser_hip_tensor = True

    def __init__(self, shape, strides, offset, storage, dtype):
        if len(shape) > LASER_MAXRANK:
            raise ValueError(f"rank {len(shape)} > LASER_MAXRANK ({LASER_MAXRANK})")
        self.shape = tuple(int(s) for s in shape)
        self.strides = tuple(int(s) for s in strides)
        self.offset = int(offset)
        self.storage = storage
        self.dtype = np.dtype(dtype)
        if self.dtype.name not in _DTYPES:
            raise TypeError(f"unsupported element type {self.dtype} ({', '.join(_DTYPES)})")

    # -- datatypes.nim:32-52 --
    @property
    def rank(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def is_C_contiguous(self):
        cur = 1
        for i in range(self.rank - 1, -1, -1):
            if self.shape[i] != 1 and self.strides[i] != cur:
                return False
            cur *= self.shape[i]
        return True

    def unsafe_raw_data(self):
        """Device address of element [offset] (the pointer can outlive the tensor: same caveat as the reference)."""
        return self.storage.raw_buffer + self.offset * self.dtype.itemsize

    # -- views --
    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if len(idx) > self.rank:
            raise IndexError("too many indices")
        shape, strides, offset = [], [], self.offset
        for d in range(self.rank):
            if d >= len(idx):
                shape.append(self.shape[d]); strides.append(self.strides[d])
                continue
            ix = idx[d]
            if isinstance(ix, slice):
                start, stop, step = ix.indices(self.shape[d])
                n = len(range(start, stop, step))
                offset += start * self.strides[d]
                shape.append(n); strides.append(self.strides[d] * step)
            else:
                i = int(ix)
                if i < 0:
                    i += self.shape[d]
                if not 0 <= i < self.shape[d]:
                    raise IndexError("index out of bounds")
                offset += i * self.strides[d]
        return Tensor(shape, strides, offset, self.storage, self.dtype)

    def transpose(self, *axes):
        axes = tuple(range(self.rank - 1, -1, -1)) if not axes else tuple(axes[0] if len(axes) == 1 and not isinstance(axes[0], int) else axes)
        if sorted(axes) != list(range(self.rank)):
            raise ValueError("axes must be a permutation")
        return Tensor([self.shape[a] for a in axes], [self.strides[a] for a in axes], self.offset, self.storage, self.dtype)

    @property
    def T(self):
        return self.transpose()

    # -- leaving the device --
    def to_numpy(self):
        t = self if self.is_C_contiguous() else deepCopy(self)
        out = np.empty(self.shape, self.dtype)
        if out.size:
            # ordered after the producing kernel / deepCopy on the current stream, complete on return
            _lib.check(_lib.lib().laser_hip_storage_download_stream(out.ctypes.data_as(C.c_void_p), C.c_void_p(t.unsafe_raw_data()),
                                                                    out.nbytes, _stream()))
        return out

    @property
    def __cuda_array_interface__(self):  # zero-copy hand-off to torch / cupy (torch.as_tensor(t, device="cuda"))
        return {"shape": self.shape, "typestr": self.dtype.str, "version": 3,
                "data": (self.unsafe_raw_data() if self.size else 0, False),
                "strides": None if self.is_C_contiguous() else tuple(s * self.dtype.itemsize for s in self.strides)}

    def __repr__(self):
        return f"Tensor[{self.dtype.name}](shape={self.shape}, strides={self.strides}, offset={self.offset}, device)"


# ---- initialization.nim ---------------------------------------------------------------------------
def newTensor(dtype, *shape):
    """newTensor[T](shape): row-major, zero-initialised (initialization.nim:156-166)."""
    if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
        shape = tuple(shape[0])
    shape = tuple(int(s) for s in shape)
    if any(s < 0 for s in shape):
        raise ValueError("negative extent")
    strides, size = _row_major_strides(shape)
    dt = np.dtype(dtype)
    return Tensor(shape, strides, 0, HipStorage(size * dt.itemsize), dt)  # storage_alloc zero-fills (allocShared0)


def toTensor(a, dtype=None):
    """toTensor(openarray): nested sequences / arrays -> a Tensor of the same shape (initialization.nim:168-202).
    Ragged nesting raises, like the reference's IndexError."""
    try:
        arr = np.array(a, dtype=dtype)
    except ValueError as e:  # inhomogeneous nesting
        raise IndexError("Each nested sequence at the same level must have the same number of elements") from e
    if arr.dtype == object:
        raise IndexError("Each nested sequence at the same level must have the same number of elements")
    if dtype is None and arr.dtype.name not in _DTYPES:
        arr = arr.astype(np.int64 if arr.dtype.kind in "iub" else np.float64)
    t = newTensor(arr.dtype, *arr.shape)
    copyFromRaw(t, np.ascontiguousarray(arr).reshape(-1), arr.size)
    return t


def fromTorch(x):
    """Non-owning Tensor view of a torch CUDA tensor (shares memory; keeps the torch tensor alive)."""
    if not x.is_cuda:
        raise TypeError("fromTorch needs a CUDA tensor")
    dt = np.dtype(str(x.dtype).replace("torch.", ""))
    st = HipStorage.external(x.data_ptr(), x.untyped_storage().nbytes() if hasattr(x, "untyped_storage") else 0, keepalive=x)
    return Tensor(tuple(x.shape), tuple(x.stride()), 0, st, dt)


def _copy_strided(dst, src):
    L = _lib.lib()
    fn = L.laser_hip_copy_strided_b32_dev if dst.dtype.itemsize == 4 else L.laser_hip_copy_strided_b64_dev
    r = dst.rank
    arr = lambda v: (C.c_int64 * max(r, 1))(*v)
    _lib.check(fn(C.c_void_p(dst.unsafe_raw_data()), arr(dst.strides), C.c_void_p(src.unsafe_raw_data()), arr(src.strides),
                  arr(dst.shape), r, _stream()))


def deepCopy(dst_or_src, src=None):
    """deepCopy(dst, src): dst gets fresh row-major storage holding src's data (dst's old storage is detached,
    never written -- initialization.nim:42-75).  With one argument, returns the copy."""
    if src is None:
        src, dst = dst_or_src, None
    else:
        dst = dst_or_src
    fresh = newTensor(src.dtype, *src.shape)
    if src.size:
        _copy_strided(fresh, src)
    if dst is None:
        return fresh
    dst.shape, dst.strides, dst.offset, dst.storage, dst.dtype = fresh.shape, fresh.strides, 0, fresh.storage, fresh.dtype
    return dst


def copyFrom(dst, src):
    """copyFrom(dst, src): same shape; only the data the destination VIEW exposes is overwritten
    (initialization.nim:77-110).  (The reference's contiguous-source fast path memcpys from the start of
    both raw buffers, ignoring offset and the destination's strides; this follows the documented contract.)"""
    if dst.shape != src.shape:
        raise ValueError(f"copyFrom: shapes differ ({dst.shape} vs {src.shape})")
    if dst.dtype != src.dtype:
        raise TypeError("copyFrom: element types differ")
    if dst.size:
        _copy_strided(dst, src)
    return dst


def copyFromRaw(dst, buffer, length):
    """copyFromRaw(dst, buffer, len): host buffer -> the destination's storage, sizes must match
    (initialization.nim:112-128, doAssert)."""
    if dst.size != int(length):
        raise AssertionError("Tensor size and buffer length should be the same")
    buf = np.ascontiguousarray(buffer, dtype=dst.dtype).reshape(-1)
    if buf.size != dst.size:
        raise AssertionError("Tensor size and buffer length should be the same")
    if not dst.size:
        return dst
    if dst.is_C_contiguous():
        _lib.check(_lib.lib().laser_hip_storage_upload_stream(C.c_void_p(dst.unsafe_raw_data()), buf.ctypes.data_as(C.c_void_p), buf.nbytes, _stream()))
    else:  # a view: stage contiguously, then scatter through the strides
        tmp = newTensor(dst.dtype, *dst.shape)
        _lib.check(_lib.lib().laser_hip_storage_upload_stream(C.c_void_p(tmp.unsafe_raw_data()), buf.ctypes.data_as(C.c_void_p), buf.nbytes, _stream()))
        _copy_strided(dst, tmp)
    return dst


def setZero(t, check_contiguous=True):
    """setZero(t): binary zero over the tensor's data; metadata untouched; contiguous input required
    (ValueError otherwise -- initialization.nim:130-154)."""
    if check_contiguous and not t.is_C_contiguous():
        raise ValueError("Input tensor is not contiguous.")
    if t.size:
        _lib.check(_lib.lib().laser_hip_storage_set_zero(C.c_void_p(t.unsafe_raw_data()), t.size * t.dtype.itemsize, _stream()))
    return t


def trimStorageCache():
    """Release the device blocks the library keeps for reuse after tensors were freed (laser_hip_storage_trim)."""
    _lib.check(_lib.lib().laser_hip_storage_trim())


__all__ = ["trimStorageCache", "Tensor", "HipStorage", "newTensor", "toTensor", "fromTorch", "deepCopy", "copyFrom", "copyFromRaw", "setZero",
           "LASER_MAXRANK", "LASER_MEM_ALIGN"]


# ---- elementwise maps: the device twin of forEach ---------------------------------------------------
# Laser's forEach (laser/strided_iteration/foreach.nim:192-264) walks raw HOST pointers; a Tensor whose storage
# lives in HBM goes through laser_hip_map_strided_*_dev instead (map_strided.hip): dst[idx] = f(a[idx] [, b[idx]])
# over strided rank <= 6 views, strides of 0 broadcast.
MAP_OPS = {"copy": 0, "fill": 1, "neg": 2, "abs": 3, "relu": 4, "scale": 5, "square": 6, "add": 32, "sub": 33, "mul": 34,
           "max": 36, "min": 37, "axpy": 38, "axpby": 39}
_MAP_SFX = {"float32": "f32", "float64": "f64", "int32": "i32", "int64": "i64"}


def _bcast_strides(t, shape):
    """Element strides of t viewed with `shape` (numpy broadcasting: missing / extent-1 dimensions get stride 0)."""
    pad = len(shape) - t.rank
    if pad < 0:
        raise ValueError("operand has more dimensions than the destination")
    out = []
    for d, n in enumerate(shape):
        if d < pad:
            out.append(0)
        else:
            e, st = t.shape[d - pad], t.strides[d - pad]
            if e == n:
                out.append(st if n != 1 else 0)
            elif e == 1:
                out.append(0)
            else:
                raise ValueError(f"operand shape {tuple(t.shape)} does not broadcast to {tuple(shape)}")
    return out


def forEachMap(op, dst, a=None, b=None, alpha=1.0, beta=0.0):
    """dst[idx] = op(a[idx] [, b[idx]]) for every index of dst -- `forEach x in dst, y in a, z in b: x = f(y, z)` for
    tensors with DEVICE storage.  op: a key of MAP_OPS; a / b broadcast against dst; dst may alias a or b."""
    code = MAP_OPS[op] if isinstance(op, str) else int(op)
    binary = code >= 32
    sfx = _MAP_SFX[dst.dtype.name]
    _ct = _lib.ctype_of(sfx)    # alpha / beta are of the element type (integers exact over the whole range)
    ct = (lambda v: _ct(int(v))) if sfx[0] == "i" else (lambda v: _ct(float(v)))
    for x in (a, b):
        if x is not None and x.dtype != dst.dtype:
            raise TypeError("operands must share the destination's element type")
    if binary and (a is None or b is None):
        raise ValueError(f"{op} needs two operands")
    if not binary and code != MAP_OPS["fill"] and a is None:
        raise ValueError(f"{op} needs an operand")
    r = dst.rank
    if r > LASER_MAXRANK:
        raise ValueError("rank > LASER_MAXRANK")
    arr = lambda v: (C.c_int64 * max(r, 1))(*v)
    L = _lib.lib()
    pa = C.c_void_p(a.unsafe_raw_data()) if a is not None else None
    sa = arr(_bcast_strides(a, dst.shape)) if a is not None else arr([0] * r)
    if binary:
        fn = getattr(L, f"laser_hip_map_strided_binary_{sfx}_dev")
        _lib.check(fn(code, C.c_void_p(dst.unsafe_raw_data()), arr(dst.strides), pa, sa, C.c_void_p(b.unsafe_raw_data()),
                      arr(_bcast_strides(b, dst.shape)), arr(dst.shape), r, ct(alpha), ct(beta), _stream()))
    else:
        fn = getattr(L, f"laser_hip_map_strided_unary_{sfx}_dev")
        _lib.check(fn(code, C.c_void_p(dst.unsafe_raw_data()), arr(dst.strides), pa, sa, arr(dst.shape), r, ct(alpha),
                      ct(beta), _stream()))
    return dst
