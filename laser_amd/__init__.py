"""laser_amd -- MI355X (gfx950) implementation of mratsim/laser's packed-panel GEMM hot path.

The product is liblaser_hip.so (hand-written HIP behind the C-ABI of include/laser_hip.h); this
package is the thin host-side mirror of the reference's Nim procs over that ABI.  No CPU fallback.
"""
from ._lib import LaserHipError, LIB_PATH, lib  # noqa: F401
from .primitives import *  # noqa: F401,F403
from . import primitives  # noqa: F401
from . import tensor  # noqa: F401
from . import sharded  # noqa: F401
from .sharded import (shard_plan, set_shard_devices, get_shard_devices, gemm_strided_sharded, matmul_sharded,  # noqa: F401
                      gemm_strided_sharded_dev, shard_rows, GATHER_NONE, GATHER_PEER, GATHER_RCCL, SHARD_PIN_TILE)
from .tensor import (Tensor, HipStorage, newTensor, toTensor, fromTorch, deepCopy, copyFrom, copyFromRaw,  # noqa: F401
                     setZero, forEachMap, MAP_OPS)

__version__ = "0.1.0"
