"""Row-panel sharded GEMM over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI), ONE collective -- an all-gather of C.

Why this shape: Laser itself partitions M across threads with no
cross-thread reduction (gemm.nim:160-176), so output rows are independent units.  Each rank owns
row panels of A, all of B (replicated) and produces the matching row panels of C; no K split, hence
no reduction and per-element arithmetic identical to the single-GPU result (parity unchanged).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so gathering 7/8 of a 2 GiB C costs
milliseconds -- comparable to the 8192^3 compute itself.  The exchange is therefore pipelined:
rows are dealt out block-cyclically in `panels_per_rank` sub-panels per rank; as soon as sub-panel s
is computed its all-gather is issued (async, RCCL's own stream) while sub-panel s+1 computes.  With
the block-cyclic deal, "sub-panel s of every rank" is one CONTIGUOUS slab of C, so each step is a
plain in-place all_gather_into_tensor straight into the final row-major C -- no staging copies.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


def _default_local_gemm(A, B, out):
    from . import primitives
    primitives.matmul(A, B, 1, 0, out)   # device-resident entry point; raises without the HIP library/GPU


# While a panel multiplies, RCCL's all-gather of the previous panel is running on the same GPU: its persistent
# workgroups (one per channel) hold some CUs for the whole transfer, and a 256x128-tile workgroup (144 KiB LDS,
# 2 x 234 VGPRs per SIMD) cannot share a CU with them.  A panel of 512 such tiles is exactly two rounds on 256 CUs but
# three on the ~224-240 left over; 128x128 tiles (2048 per panel, 2 per CU) degrade gracefully instead (about -3 % when
# all CUs are free).  Multi-GPU runs therefore pin that tile CLASS of the hand-scheduled assembly kernels (option
# "asm_tile" = 2: lh_f32_exact_128x128x16 / lh_f32_fast_128x128x16 by accumulation mode; round 5 -- the pin used to
# force the compiler-scheduled 128x128 configuration, 10-15 % slower per GPU) for the local products unless told otherwise.
SHARDED_ASM_TILE = 2
SHARDED_TILE_NAME = "lh_f32_*_128x128x16 (hand-scheduled assembly, option asm_tile=2)"


@dataclass
class PanelPlan:
    M: int
    world: int
    panels_per_rank: int
    rows: int           # rows per sub-panel (last ones may be ragged / empty)

    @property
    def padded_M(self):
        return self.rows * self.world * self.panels_per_rank

    def slab(self, s):
        """Row range of C that step s's all-gather fills (sub-panel s of every rank)."""
        return s * self.world * self.rows, (s + 1) * self.world * self.rows

    def panel(self, s, rank):
        """(row_start, valid_rows) of rank's sub-panel s."""
        start = (s * self.world + rank) * self.rows
        return start, max(0, min(self.rows, self.M - start))


def make_plan(M, world, panels_per_rank=4, align=256):
    """Block-cyclic deal of M rows: world*panels_per_rank panels of `rows` rows, rows a multiple of
    `align` (the largest workgroup tile) when M allows it so every panel runs the vector loaders."""
    panels_per_rank = max(1, int(panels_per_rank))
    n = world * panels_per_rank
    rows = -(-M // n)
    if rows >= align:
        rows = -(-rows // align) * align
    while panels_per_rank > 1 and rows * world * (panels_per_rank - 1) >= M:
        panels_per_rank -= 1   # do not create all-empty steps
    return PanelPlan(M, world, panels_per_rank, rows)


class ShardedGemm:
    """C[M,N] = A[M,K] . B[K,N] with A's row panels dealt over the ranks of `group`."""

    def __init__(self, M, N, K, dtype=torch.float32, device=None, group=None, panels_per_rank=4,
                 local_gemm=None, tile_config=None, gather="collective"):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.M, self.N, self.K = M, N, K
        self.dtype, self.device = dtype, device
        self.plan = make_plan(M, self.world, panels_per_rank)
        self.local_gemm = local_gemm or _default_local_gemm
        # fp32 tile for the local products: None = the SHARDED_ASM_TILE class of the assembly kernels when more than one rank
        # shares the work (single rank: the library heuristic); -1 = the library heuristic; >= 0 = that compiler-scheduled
        # configuration of laser_amd.f32_configs() (tuning sweeps)
        self.tile_config = tile_config
        self._pin_tiles = local_gemm is None and dtype == torch.float32
        # how sub-panel s reaches the other ranks: "collective" = one in-place all_gather_into_tensor per slab;
        # "p2p" = the same exchange as world-1 grouped send/recv pairs (ncclGroup of point-to-point operations): every
        # rank pushes its rows straight to each peer, which on a fully connected xGMI node puts all 7 links to work
        # whatever ring / tree the collective's algorithm picks
        if gather not in ("collective", "p2p"):
            raise ValueError("gather is 'collective' or 'p2p'")
        self.gather = gather

    # -- data placement helpers ------------------------------------------------------------------
    def local_rows(self):
        """Global row indices this rank owns, in local storage order."""
        idx = []
        for s in range(self.plan.panels_per_rank):
            start, valid = self.plan.panel(s, self.rank)
            idx.extend(range(start, start + valid))
        return idx

    def shard_A(self, A_full):
        """[M,K] -> this rank's panels, stacked [panels_per_rank*rows, K] (ragged rows zero)."""
        p = self.plan
        out = torch.zeros((p.panels_per_rank * p.rows, self.K), dtype=A_full.dtype, device=A_full.device)
        for s in range(p.panels_per_rank):
            start, valid = p.panel(s, self.rank)
            if valid > 0:
                out[s * p.rows: s * p.rows + valid] = A_full[start:start + valid]
        return out

    def alloc_C(self):
        """Full C, row-major, padded to whole panels; use result()[..] to view the M valid rows."""
        return torch.zeros((self.plan.padded_M, self.N), dtype=self.dtype, device=self.device)

    # -- the hot path -------------------------------------------------------------------------------
    def run(self, A_local, B, C_full):
        """A_local: [panels_per_rank*rows, K] (from shard_A); B: [K,N] replicated; C_full: alloc_C().
        On return (after the stream/work sync at the end) every rank holds all of C."""
        cfg = self.tile_config
        prev_cfg = prev_tile = primitives = None
        if self._pin_tiles:
            from . import primitives
        if self._pin_tiles and cfg is None and self.world > 1:
            # per THREAD (the process-wide option forced every other thread's f32 GEMMs onto this tile class for the
            # duration of the run, and stayed set if the process died inside it); the caller's own setting is restored
            prev_tile = primitives.get_option("thread_asm_tile")
            primitives.set_option("thread_asm_tile", SHARDED_ASM_TILE)
        elif self._pin_tiles and cfg is not None:
            prev_cfg = primitives.get_f32_config()
            primitives.set_f32_config(cfg)
        try:
            works = self._run_panels(A_local, B, C_full)
        finally:
            if prev_tile is not None:
                primitives.set_option("thread_asm_tile", prev_tile)
            if prev_cfg is not None:
                primitives.set_f32_config(prev_cfg)
        for w in works:
            w.wait()
        return C_full[: self.M]

    def _run_panels(self, A_local, B, C_full):
        p = self.plan
        works = []
        for s in range(p.panels_per_rank):
            start, valid = p.panel(s, self.rank)
            if valid > 0:
                self.local_gemm(A_local[s * p.rows: s * p.rows + valid], B, C_full[start:start + valid])
            if self.world > 1:
                lo, hi = p.slab(s)
                mine = C_full[start:start + p.rows]          # in-place: my slice of the slab
                if self.gather == "p2p":
                    works.extend(_p2p_gather_rows(C_full[lo:hi], mine, self.rank, self.world, self.group))
                else:
                    works.append(_all_gather_rows(C_full[lo:hi], mine, self.group))
        return works


def _all_gather_rows(slab, mine, group):
    """In-place all-gather of equal row blocks into the contiguous `slab`."""
    try:
        return dist.all_gather_into_tensor(slab, mine, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):
        # backends without a flat all-gather (older gloo): list form over views of the same slab
        world = dist.get_world_size(group)
        rows = mine.shape[0]
        outs = [slab[r * rows:(r + 1) * rows] for r in range(world)]
        return dist.all_gather(outs, mine.clone(), group=group, async_op=True)


def _p2p_gather_rows(slab, mine, rank, world, group):
    """The same exchange as grouped point-to-point operations: send my rows to every peer, receive every peer's rows into
    its slot of the slab (in place).  Returns the list of work handles."""
    rows = mine.shape[0]
    ops = []
    for d in range(1, world):          # rotate the peer order per rank so no single rank is everyone's first target
        peer = (rank + d) % world
        src = (rank - d) % world
        ops.append(dist.P2POp(dist.isend, mine, peer, group))
        ops.append(dist.P2POp(dist.irecv, slab[src * rows:(src + 1) * rows], src, group))
    return dist.batch_isend_irecv(ops) if ops else []

