"""Host-side mirror of the single-process sharded entry points of liblaser_hip.so (include/laser_hip.h, "sharded"
section; implementation laser_amd/csrc/sharded.cpp): gemm_strided cut into row panels over the GPUs of one node, one
process driving all of them.  Rows of C are independent units in Laser (gemm.nim:160-176: `omp for` over the ic row
blocks, no cross-thread reduction), so results are bit-identical whatever the device count.

The one-process-per-GPU form over torch.distributed / RCCL lives in laser_amd/distributed.py; both deal the rows with
the same block-cyclic rule (shard_plan == distributed.make_plan)."""
import ctypes as C

import numpy as np

from . import _lib
from .primitives import _sfx

GATHER_NONE, GATHER_PEER, GATHER_RCCL = 0, 1, 2
SHARD_PIN_TILE = 1


def shard_plan(M, ndev, panels_per_dev=4):
    """(rows_per_panel, panels_per_dev actually used, padded_M): panel (s, g) is rows [(s*ndev + g)*rows, +rows)."""
    rows, ppd, padded = C.c_int64(), C.c_int(), C.c_int64()
    _lib.check(_lib.lib().laser_hip_shard_plan(int(M), int(ndev), int(panels_per_dev), C.byref(rows), C.byref(ppd), C.byref(padded)))
    return rows.value, ppd.value, padded.value


def set_shard_devices(ndev):
    """Route large plain host-pointer gemm_strided calls over `ndev` GPUs (1 = off, 0 = every visible GPU)."""
    _lib.check(_lib.lib().laser_hip_set_shard_devices(int(ndev)))


def get_shard_devices():
    return _lib.lib().laser_hip_get_shard_devices()


def _devs(devices):
    if devices is None:
        return None
    return (C.c_int * len(devices))(*[int(d) for d in devices])


def gemm_strided_sharded(devices, M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C_,
                         rowStrideC, colStrideC, ndev=None):
    """Host pointers (numpy): gemm_strided's parameter list after the device list; blocks until C is final."""
    s = _sfx(C_)
    ct = _lib.ctype_of(s)
    n = len(devices) if devices is not None else int(ndev or 0)
    fn = getattr(_lib.lib(), f"laser_hip_gemm_strided_{s}_sharded")
    _lib.check(fn(n, _devs(devices), M, N, K, ct(alpha), C.c_void_p(A.ctypes.data), rowStrideA, colStrideA,
                  C.c_void_p(B.ctypes.data), rowStrideB, colStrideB, ct(beta), C.c_void_p(C_.ctypes.data), rowStrideC, colStrideC))
    return C_


def matmul_sharded(A, B, devices, alpha=1, beta=0, out=None):
    """numpy convenience over gemm_strided_sharded for 2-D views of any strides."""
    M, K = A.shape
    _, N = B.shape
    if out is None:
        out = np.zeros((M, N), dtype=A.dtype)
    es = lambda x: tuple(st // x.dtype.itemsize for st in x.strides)
    (rsA, csA), (rsB, csB), (rsC, csC) = es(A), es(B), es(out)
    return gemm_strided_sharded(devices, M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, out, rsC, csC)


def gemm_strided_sharded_dev(devices, M, N, K, alpha, A_panels, rowStrideA, colStrideA, Bs, rowStrideB, colStrideB, beta,
                             Cs, rowStrideC, panels_per_dev=4, gather=GATHER_PEER, flags=0):
    """Device-resident: A_panels[g] / Bs[g] / Cs[g] are torch tensors on devices[g] (device g's stacked row panels of A,
    a replica of B, the full row-major C).  Blocks until every device holds all of C (gather != GATHER_NONE)."""
    s = _sfx(Cs[0])
    ct = _lib.ctype_of(s)
    n = len(devices)
    if not (len(A_panels) == len(Bs) == len(Cs) == n):
        raise ValueError("one A panel stack, one B and one C per device slot")
    tab = lambda xs: (C.c_void_p * n)(*[C.c_void_p(x.data_ptr()) for x in xs])
    # The C entry point has no stream parameter: it works on its own (non-blocking) streams, one set per device slot, and its
    # contract is that the operands are READY when it is called.  torch fills tensors asynchronously on its current stream, so the
    # producers of every operand are waited for here (round 5: a panel made by shard_rows a moment before the call was read
    # half-written by the slot's first product -- tests/test_gpu_sharded.py caught it with the pinned one-chain kernel).
    try:
        import torch
        for d in sorted({x.device.index for x in list(A_panels) + list(Bs) + list(Cs) if isinstance(x, torch.Tensor) and x.is_cuda}):
            torch.cuda.current_stream(d).synchronize()
    except ImportError:
        pass
    fn = getattr(_lib.lib(), f"laser_hip_gemm_strided_{s}_sharded_dev")
    _lib.check(fn(n, _devs(devices), M, N, K, ct(alpha), tab(A_panels), rowStrideA, colStrideA, tab(Bs), rowStrideB, colStrideB,
                  ct(beta), tab(Cs), rowStrideC, int(panels_per_dev), int(gather), int(flags)))
    return Cs


def shard_rows(A_full, ndev, g, panels_per_dev=4):
    """[M, K] torch/numpy matrix -> device slot g's panels stacked [ppd*rows, K] (ragged rows zero), the layout
    gemm_strided_sharded_dev expects in A_panels[g]."""
    M = A_full.shape[0]
    rows, ppd, _ = shard_plan(M, ndev, panels_per_dev)
    import torch
    out = torch.zeros((ppd * rows, A_full.shape[1]), dtype=A_full.dtype, device=A_full.device) if isinstance(A_full, torch.Tensor) \
        else np.zeros((ppd * rows, A_full.shape[1]), dtype=A_full.dtype)
    for s_ in range(ppd):
        start = (s_ * ndev + g) * rows
        valid = max(0, min(rows, M - start))
        if valid > 0:
            out[s_ * rows: s_ * rows + valid] = A_full[start:start + valid]
    return out
