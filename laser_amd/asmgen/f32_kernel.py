"""Generator of the hand-scheduled fp32 GEMM kernels (gfx950 assembly), the MI355X twin of the reference's generated
micro-kernels (laser/primitives/matrix_multiplication/gemm_ukernel_generator.nim:140-250: every load, broadcast and FMA
placed by the author) and of its packing + loop nest (gemm_packing.nim:24-94, gemm.nim:109-176).

Shape of the kernel (one workgroup = one BM x BN tile of C, 4 waves = ONE wave per SIMD, 512 registers per lane):
  * wave tile (BM/2) x (BN/2) of 32x32 v_mfma_f32_32x32x2_f32 blocks, accumulators in AGPRs; in laser-order mode a second
    AGPR set holds the running sum of Laser's kc = 512 slices (gemm.nim:150-158: every slice is a chain from +0, the
    slice sums are added in order) -- bit-identical to the reference on an FMA host;
  * operand tiles HBM -> registers (16-byte buffer loads, bounds-checked descriptors: anything past the operand reads as
    0) -> LDS in the k-quad panel image (one ds_read_b128 hands a lane four k-steps of a fragment) -- this
    IS Laser's packing stage; 3-stage LDS ring, one s_barrier per K-tile placed where every wave has an MFMA in flight;
  * the K-tile body is one straight-line block: 128 MFMAs with every ds_read / ds_write / buffer_load / SALU op assigned
    to a fixed gap between two MFMAs and every s_waitcnt counted by the generator from its own model of the two
    memory queues (never 0 for vmcnt in the steady state: tile t+2 is in flight while tile t multiplies).
Restrictions (the launcher falls back to the compiler-scheduled kernels otherwise): float32, A and C row-major-like
(unit column stride), B unit column stride (NN) or unit row stride (NT), K a multiple of BK, alpha == 1, beta == 0.
"""
from .core import Prog, Reg, Sym, v, a, s, VCC, EXEC, OFF, M0


class Cfg:
    def __init__(self, name, BM, BN, BK, exact, bar_gap=None, w_start=2, w_step=None, trace=False, b_kcontig=False,
                 b_store="write2", ablate=(), r_step=1, debug=False, filler=None, filler_every=1, conv=False, dtype="f32",
                 persistent=None, pre=False, deep=False, pipe=None, il=False, runv=False, dataa=None, cpers=False):
        self.name, self.BM, self.BN, self.BK, self.exact = name, BM, BN, BK, exact
        # cpers (round 6): a convolution kernel whose workgroups walk units (image, tile) g, g + G, g + 2G ... of the launch with
        # PIPELINED transitions (pipe, below) -- a scheduler of three SGPRs (next unit, image, armed bit): the K-slice hand-overs of the
        # GEMM scheduler do not exist here (the convolution is never cut along K), and its state would not fit beside the tap arithmetic
        self.cpers = cpers
        assert not cpers or (conv and not debug and not persistent)
        if cpers:
            persistent, pipe = False, True
        # il (round 6): the three LDS stages are INTERLEAVED by row -- row r of stage s at r * 3 * RS + s * RS -- instead of three
        # consecutive panel images.  A stage is then an immediate offset of the LDS instructions (s * 128 bytes: inside the 16-bit offset
        # of ds_read_b128 next to the block offset, and inside ds_write2_b32's 8-bit dword offsets), so ONE address register per
        # (group / piece) serves all three stages where the consecutive layout needs three: 96 address registers become 32.  The bank
        # of every access is unchanged (the row pitch 96 dwords = 32 mod 64, the pitch of the consecutive layout).
        # runv (needs il): the laser-order running sum lives in arch VGPRs (the registers il frees), the fragment and staging registers
        # move to AGPRs (LDS and buffer instructions take either file, and so do the MFMA's A / B operands): the slice fold is then
        # v_accvgpr_read + v_pk_add_f32 -- 1.5 VALU operations per element where the all-AGPR plan needs 4.
        self.il, self.runv = il, runv
        self.dataa = (runv and il) if dataa is None else dataa        # fragments + staging in AGPRs (where the running sum would not fit beside them)
        assert not (runv and not exact) and not (il and (deep or dtype != "f32" or b_store != "write2"))
        assert not (self.dataa and pre), "the fused prologue works on the staging registers with VALU instructions"
        # deep (generator option, no shipped kernel uses it): TWO sets of staging registers -- a K-tile is requested two tile bodies before
        # it is stored to LDS instead of one; six tile bodies instead of three (LDS stage x register set).  Built for the small tile,
        # whose body is 16 MFMAs per wave and which runs 0.965 of peak with its global loads ablated against 0.83 - 0.87 with them:
        # measured, it changes nothing (3072^3: 135.5 / 135.1 / 136.0 against 131.8 / 135.6 / 136.7 TFLOP/s, interleaved; 1024^3 110.3
        # against 110.9: profiles/r04/asm_probe_64x64_deep_*.jsonl) -- the loads cost bandwidth there, not latency.
        self.deep = deep
        assert not (deep and (conv or pre or debug))
        # pre: fused prologue (README.md:243-244: "fuse operations before the matrix multiplication kernel, during the prepacking"):
        # relu applied to the elements of A and / or B in the staging registers, on their way into the LDS panel image; which operand
        # is a run-time mask (KA_PRE).  Kernels of their own: the VALU work costs matrix-pipe time, the plain kernels carry none of it.
        self.pre = pre
        if pre and persistent is None:
            persistent = False          # (the scheduler state and the masks do not both fit the SGPR file)
        self.dtype = dtype
        # persistent: the workgroup loops over unit runs handed out by the in-kernel scheduler (sched_next): whole tiles, and -- where
        # the launcher cuts tiles along K at slice boundaries -- head slices stored to a workspace / tail runs that fold them in
        # order.  The convolution kernels (out of SGPRs, never split) run exactly one tile per workgroup.
        self.persistent = (not conv and not debug) if persistent is None else persistent
        # pipe (round 6): tile transitions of a persistent workgroup are software-pipelined.  When the run that follows a whole tile is
        # another whole tile, the K loop never stops: the last two tile bodies of tile T already fetch the first two K-tiles of tile
        # T + 1 (today they fetch zeros past K), and the first body of tile T + 1 is a TRANSITION body -- a fold tile whose fold also
        # finishes tile T: per block, the chain set is read out, the block's first MFMA restarts the chain from +0 for the new tile,
        # and C = run + alpha * slice leaves for memory from the gap behind that MFMA (the running-sum set is zeroed on the way).  No
        # drain, no prologue, no first-load latency, no burst of C stores between two tiles of a workgroup.
        self.pipe = (self.persistent and dtype == "f32" and not conv and not pre and not deep and not debug) if pipe is None else pipe      # (f32x16: per configuration, f32x16_kernel.py)
        assert not (self.pipe and ((not self.persistent and not cpers) or deep or debug or pre))
        f64, x16 = dtype == "f64", dtype == "f32x16"
        # f32: v_mfma_f32_32x32x2 (32x32 blocks, 2 k per instruction, one 16-byte fragment read feeds 4 k-steps);
        # f64: v_mfma_f64_16x16x4 (16x16 blocks, 4 k per instruction, one 16-byte read feeds 2 k-steps): 8 k per group either way;
        # f32x16 (f32x16_kernel.py): v_mfma_f32_16x16x4 (16x16 blocks, 4 k per instruction, one 16-byte read feeds 4 k-steps: 16 k per group)
        self.MB, self.KSTEP, self.KREAD, self.ACCR, self.ESZ = (16, 4, 2, 8, 8) if f64 else (16, 4, 4, 4, 4) if x16 else (32, 2, 4, 16, 4)
        self.KC = 256 if f64 else 512    # gemm_tiling.nim:310: kc = 2048 / sizeof(T)
        # bytes of one panel row in LDS (f64 / f32x16: 128 + 16 of padding: conflict-free b128 reads without a swizzle)
        self.RS = 144 if f64 else (BK * 4 + 16) if x16 else BK * 4
        self.WTM, self.WTN = BM // 2, BN // 2
        self.TM, self.TN = self.WTM // self.MB, self.WTN // self.MB
        self.NB = self.TM * self.TN
        self.NG = BK // (16 if x16 else 8)            # fragment groups (8 k each; f32x16: 16 k) per K-tile
        self.NMF = self.NB * (BK // self.KSTEP)  # MFMAs per K-tile per wave
        self.GM = self.NMF // self.NG   # MFMAs per group
        self.STAGE = (BM + BN) * self.RS
        self.ROWP = 3 * self.RS if il else self.RS      # bytes between two rows of a panel in LDS
        self.NPA = BM * BK * self.ESZ // 16 // 256   # 16-byte pieces of A per thread per tile
        self.NPB = BN * BK * self.ESZ // 16 // 256
        # implicit-GEMM convolution (round 6: any kH x kW of up to `ntmax` taps, strides 1 .. 255, any zero padding, any output
        # width -- conv2d_im2col.nim:42-88 is generic in all of them): B is the NCHW image; a piece = 2 output pixels per lane of one
        # k = (channel, kernel row, kernel column), gathered by two dword loads whose per-lane offsets come from a table in LDS (one
        # entry per kernel tap + "nothing"), indexed by the wave-uniform tap.  The table is sized at generation time: 31 taps beside
        # the 3 x 48 KiB stages of the 256-row tile (5 x 5, 3 x 7 ...), 49 (7 x 7) beside the smaller tiles
        self.conv = conv
        self.ntmax = (31 if BM >= 256 else 49) if conv else 0
        self.TAB_ENTRY, self.TAB_E1 = 256, 256 * (self.ntmax + 1)   # bytes per table entry (64 lanes x 4), offset of the second pixel's table
        self.LDS0 = 2 * self.TAB_E1 if conv else 0       # the table sits below the stage ring (ds_read_addtid reaches 64 KiB)
        if conv:
            assert (BN, BK) == (128, 32) and not b_kcontig
            self.NPB = 8
        assert self.NPB % 2 == 0 or b_kcontig or dtype != "f32"
        self.KC_TILES = self.KC // BK
        self.bar_gap = bar_gap if bar_gap is not None else (self.NMF - self.GM - 1)
        self.w_start, self.w_step = w_start, w_step
        self.trace = trace
        self.b_kcontig = b_kcontig
        self.b_store = b_store      # "write2": ds_write2_b32 straight from the two pieces; "swap64": 4 v_swap + 4 ds_write_b64
        self.ablate = set(ablate)   # timing experiments only (results are wrong): "loads", "stores", "reads", "barrier", "cstores" (the epilogue's stores of C)
        self.r_step = r_step
        self.filler, self.filler_every = filler, filler_every   # pricing experiments: one extra instruction of this kind per gap
        self.debug = debug          # dump intermediate state of workgroup 0 to the kernarg's debug buffer (asm_debug.py)
        self.lds_bytes = self.LDS0 + 3 * self.STAGE
        assert self.lds_bytes <= 160 * 1024
        # persistent laser-order kernels: 16 bytes past the stage ring where the four waves agree on what they saw in a flag
        self.VOTE = self.lds_bytes
        self.lds_alloc = self.lds_bytes + (16384 if filler else 0) + (16 if (self.persistent and exact) else 0)
        assert self.lds_alloc <= 160 * 1024


CONFIGS = {
    # barrier positions from the schedule sweeps (profiles/r03/asm_probe_v1.jsonl, asm_probe_v2.jsonl)
    "exact_256x128x32": dict(BM=256, BN=128, BK=32, exact=True, bar_gap=95, w_step=1, il=True, runv=True),
    "fast_256x256x16": dict(BM=256, BN=256, BK=16, exact=False, bar_gap=95),
    # mid-size problems (fewer than one round of the large tiles): 128 VGPRs + 128 AGPRs and 48 KiB of LDS per workgroup, so
    # two workgroups share a CU -- two waves per SIMD that cover each other's barrier and waits
    "exact_128x128x16": dict(BM=128, BN=128, BK=16, exact=True, runv=True),
    "fast_128x128x16": dict(BM=128, BN=128, BK=16, exact=False),
    # B passed transposed (rowStrideB == 1: k-contiguous like A) -- BASELINE configs[2]
    "exact_256x128x32_nt": dict(BM=256, BN=128, BK=32, exact=True, bar_gap=95, w_step=1, b_kcontig=True, il=True, runv=True),
    "fast_256x256x16_nt": dict(BM=256, BN=256, BK=16, exact=False, bar_gap=95, b_kcontig=True),
    "exact_128x128x16_nt": dict(BM=128, BN=128, BK=16, exact=True, b_kcontig=True, runv=True),
    "fast_128x128x16_nt": dict(BM=128, BN=128, BK=16, exact=False, b_kcontig=True),
    # one chain on the laser-order kernels' tile: halves the tile quantisation of the 256x256 tile (4100^3: 289 tiles of
    # 256x256 are 1.13 rounds of the chip, 561 tiles of 256x128 are 2.19)
    # small tiles for problems with few large tiles (1024^3 = 32 tiles of 256x128 on 256 CUs) and for the tile quantisation of
    # mid-size ones: 2 - 3 workgroups share a CU
    "exact_64x64x32": dict(BM=64, BN=64, BK=32, exact=True, runv=True),
    "fast_64x64x32": dict(BM=64, BN=64, BK=32, exact=False),
    "exact_64x64x32_nt": dict(BM=64, BN=64, BK=32, exact=True, b_kcontig=True, runv=True),
    "fast_64x64x32_nt": dict(BM=64, BN=64, BK=32, exact=False, b_kcontig=True),
    # implicit-GEMM convolution, 3x3 kernel, stride 1, any zero padding (benchmarks/convolution/conv2d_im2col.nim)
    "conv_exact_256x128x32": dict(BM=256, BN=128, BK=32, exact=True, il=True, runv=True, bar_gap=95, conv=True),
    "conv_fast_256x128x32": dict(BM=256, BN=128, BK=32, exact=False, bar_gap=95, conv=True),
    # fewer output channels: 128 / 64 rows of the same pixel tile (the B side -- the gather -- is unchanged)
    "conv_exact_128x128x32": dict(BM=128, BN=128, BK=32, exact=True, il=True, runv=True, conv=True),
    "conv_fast_128x128x32": dict(BM=128, BN=128, BK=32, exact=False, conv=True),
    "conv_exact_64x128x32": dict(BM=64, BN=128, BK=32, exact=True, il=True, runv=True, conv=True),
    "conv_fast_64x128x32": dict(BM=64, BN=128, BK=32, exact=False, conv=True),
    # the convolution kernels as unit walkers (Cfg.cpers): workgroup g runs units (image, tile) g, g + G, ... with pipelined transitions
    "conv_exact_256x128x32_p": dict(BM=256, BN=128, BK=32, exact=True, il=True, runv=True, bar_gap=95, conv=True, cpers=True),
    "conv_fast_256x128x32_p": dict(BM=256, BN=128, BK=32, exact=False, bar_gap=95, conv=True, cpers=True),
    "conv_exact_128x128x32_p": dict(BM=128, BN=128, BK=32, exact=True, il=True, runv=True, conv=True, cpers=True),
    "conv_fast_128x128x32_p": dict(BM=128, BN=128, BK=32, exact=False, conv=True, cpers=True),
    "conv_exact_64x128x32_p": dict(BM=64, BN=128, BK=32, exact=True, il=True, runv=True, conv=True, cpers=True),
    "conv_fast_64x128x32_p": dict(BM=64, BN=128, BK=32, exact=False, conv=True, cpers=True),
    "fast_256x128x32": dict(BM=256, BN=128, BK=32, exact=False, bar_gap=95),
    "fast_256x128x32_nt": dict(BM=256, BN=128, BK=32, exact=False, bar_gap=95, b_kcontig=True),
    # one round of 128x128 tiles (129 .. 256 of them: 1920^3, 2048^3): a workgroup has its CU to itself, and the 16-deep K-tile of
    # the two-per-CU kernels (32 MFMAs between barriers) leaves its latencies uncovered; 32 deep = 64 MFMAs per barrier, 96 KiB of LDS
    # fused prologue (relu on A and / or B elements on their way into LDS): variants of the main tiles
    **{f"{base}_pre{nt}": dict(BM=bm, BN=bn, BK=bk, exact=ex, pre=True, **({"bar_gap": bg} if bg else {}), **({"b_kcontig": True} if nt else {}))
       for base, bm, bn, bk, ex, bg in (("exact_256x128x32", 256, 128, 32, True, 95), ("fast_256x256x16", 256, 256, 16, False, 95),
                                        ("exact_128x128x16", 128, 128, 16, True, None), ("fast_128x128x16", 128, 128, 16, False, None),
                                        ("exact_64x64x32", 64, 64, 32, True, None), ("fast_64x64x32", 64, 64, 32, False, None))
       for nt in ("", "_nt")},
    "exact_128x128x32": dict(BM=128, BN=128, BK=32, exact=True, runv=True),
    "fast_128x128x32": dict(BM=128, BN=128, BK=32, exact=False),
    "exact_128x128x32_nt": dict(BM=128, BN=128, BK=32, exact=True, b_kcontig=True, runv=True),
    "fast_128x128x32_nt": dict(BM=128, BN=128, BK=32, exact=False, b_kcontig=True),
}

# kernel argument block (bytes)
KA_A, KA_B, KA_C, KA_TAB, KA_LDA, KA_LDB, KA_LDC, KA_M, KA_N, KA_K, KA_DBG = 0, 8, 16, 24, 32, 36, 40, 44, 48, 52, 64
# convolution kernels only: H W oW pH pW Cin Npix magic(oW) | shift(oW) - bsB(bytes, u64) | bsC(bytes, u64)
KA_CONV0, KA_CONV1, KA_CONV2 = 72, 104, 120
KA_BIAS, KA_EPI = 128, 136   # fused epilogue: bias pointer (u64; 0 = none); rowStrideBias, colStrideBias (elements), activation (0 none / 1 relu); GEMM kernels: column stride of C in elements (0 = 1)
KA_BSA = 72      # GEMM kernels: batch stride of A in bytes (u64); B's and C's share the convolution kernels' slots at 112 / 120
KA_PRE = 108     # GEMM kernels with a fused prologue: bit 0 = relu on A's elements, bit 1 = relu on B's (x > 0 ? x : 0)
# scheduler block (every kernel): the workgroup -> tile map is arithmetic on these (gemm.nim:160-176 partitions by arithmetic too);
# a divisor d travels as magic(d) = floor(2^32 / d) + 1 (0 for d == 1): x / d = mulhi(x, magic) while x * d < 2^32 (launcher)
#   +0 tiles_m  +4 tiles_n  +8 group_m  +12 rows of the last group  +16 magic(group_m * tiles_n)  +20 magic(group_m)
#   +24 magic(rows of the last group)  +28 xcd_q (workgroups / 8; 0 = no XCD remap)
#   +32 xcd_r (workgroups % 8)  +36 P (K slices per tile)  +40 magic(P)  +44 units_q  +48 units_r (units = workgroups * q + r:
#       workgroup v starts at unit v * q + floor(v * r / workgroups) -- the r longer ranges are spread evenly over the ids)
#   +52 slice length (elements of K)  +56 bit 0: never take a received sum early (tests: forces the two-run receive path);
#       bit 1: two-level ranges (launches that cut tiles; workgroups a multiple of 8): XCD x = workgroup id % 8 owns the WHOLE tiles
#       [T x / 8, T (x + 1) / 8) and its workgroups (local index l = id / 8) share those tiles' units evenly -- a tile is only ever
#       cut between workgroups id and id + 8: the hardware starts the sender before the receiver, so a receiver never waits for a
#       workgroup that is not running yet, whatever else shares the GPU.  Then +44 = T (tiles) and +60 = magic(workgroups / 8)
#   +60 magic(workgroups)
#   +64 workspace (u64)  +72 flags (u64)
KA_SCHED = 152
KA_SCHED2 = KA_SCHED + 32
KA_WS = KA_SCHED + 64
KERNARG_SIZE = 232
MODE_NORMAL, MODE_ACC, MODE_CONT = 0, 1, 2     # run_setup: running sum from beta * C0 / left alone (not used by the run) / kept (a received sum)
END_EPI, END_SEND, END_RECV = 0, 1, 2          # after the K loop: store C / send the running sum on / receive the predecessor's


def magic_u32(d):
    """host side of the kernels' division: floor(2^32 / d) + 1, 0 for d == 1"""
    assert d >= 1
    return 0 if d == 1 else ((1 << 32) // d + 1) & 0xffffffff


def c_runv(gen):
    return getattr(gen.c, "runv", False)


class Gen:
    def __init__(self, cfg):
        self.c = cfg
        self.p = Prog()
        self.vmq = []   # tags of outstanding VMEM loads, oldest first
        self.lgq = []   # tags of outstanding LDS ops
        self.alloc()

    # ------------------------------------------------------------------ registers
    def alloc(self):
        c, p = self.c, self.p
        S, V = p.salloc, p.valloc
        self.sA, self.sB, self.sC, self.sTAB = None, None, None, None
        self.ka0 = S(8, align=4)   # A, B, C, TAB pointers
        self.ka1 = S(8, align=4)   # lda ldb ldc M N K - -
        self.s_lda, self.s_ldb, self.s_ldc, self.s_M, self.s_N, self.s_K = (self.ka1[i] for i in range(6))
        self.s_alpha, self.s_beta = self.ka1[6], self.ka1[7]     # float32 bit patterns
        self.srdA, self.srdB, self.srdC = S(4), S(4), S(4)
        self.s_rem, self.s_cnt = S(), S()
        self.s_bstep = S()
        self.s_m0, self.s_n0, self.s_wave, self.s_wm0, self.s_wn0 = S(), S(), S(), S(), S()
        self.s_t = [S() for _ in range(6)]
        self.s_ldc4, self.s_ldc20 = S(), S()
        self.s_csC4 = None if c.conv else S()      # column stride of C in bytes (KA_EPI + 12; 0 in the arguments = dense)
        self.s_preA, self.s_preB = (S(2, align=2), S(2, align=2)) if c.pre else (None, None)   # lane masks: all ones = apply relu
        self.alloc_sched()
        # accumulators
        self.acc = [p.aalloc(c.ACCR) for _ in range(c.NB)]
        # (runv: the running sum in arch VGPRs -- VALU instructions cannot address AGPRs -- and the data-only registers in AGPRs)
        self.run = ([V(c.ACCR) if c.runv else p.aalloc(c.ACCR) for _ in range(c.NB)]) if c.exact else None
        D4 = (lambda n: p.aalloc(n)) if c.dataa else V
        # fragments: 2 slots
        self.fa = [[D4(4) for _ in range(c.TM)] for _ in range(2)]
        self.fb = [[D4(4) for _ in range(c.TN)] for _ in range(2)]
        # staging pieces
        self.stA = [D4(4) for _ in range(c.NPA)]
        self.stB = [D4(2) for _ in range(c.NPB)] if c.conv else [D4(4) for _ in range(c.NPB)]
        # deep: tile t waits in set t & 1 (the loop body that multiplies tile t stores tile t + 1 from its set and requests tile t + 3 into it)
        self.st_sets = [(self.stA, self.stB)]
        if c.deep:
            self.st_sets.append(([V(4) for _ in range(c.NPA)], [V(4) for _ in range(c.NPB)]))
        # per-lane LDS addresses, one register per LDS stage: [0] = the stage the tile being multiplied lives in (reads) /
        # the stage being filled (writes), [1] = the next one, [2] = the third; rotated with v_swap_b32 once per K-tile --
        # v_swap is free beside the MFMA stream while any other VALU op costs ~11 cycles of matrix-pipe time
        # (profiles/r03/asm_probe_v3_fillers.jsonl, asm_probe_v4_fillers.jsonl), so the loop computes no address at all
        # (il: one register + the stage as an immediate offset: an entry is (register, bytes); see lds_at)
        def triple(read):
            if not c.il:
                return [V() for _ in range(3)]
            r = V()
            return [(r, 0), (r, c.RS), (r, 2 * c.RS)] if read else [(r, c.RS), (r, 2 * c.RS), (r, 0)]
        self.RA = [triple(True) for _ in range(c.NG)]
        self.RB = [triple(True) for _ in range(c.NG)]
        self.WA = [[triple(False) for _ in range(c.NPA)] for _ in range(2)]   # [MFMA half][piece][stage]
        if c.conv:
            self.WB = [[triple(False) for _ in range(2)] for _ in range(4)]       # [pair][pixel of the piece][stage]
            self.vB0 = [V() for _ in range(8)]             # per piece: buffer offset of the lane's first / second pixel for the
            self.vB1 = [V() for _ in range(8)]             # piece's tap (read from the LDS table; 0x80000000 = padding)
            scr = S(16, align=4)
            self.s_scr = scr
            self.s_koff = [scr[i] for i in range(8)]       # per piece: (c*H*W + kh*W + kw) * 4 of the k it gathers
            self.s_k0 = S()                                # this wave's first k (= c * taps + kh * kW + kw) in the tile being loaded
            self.s_NT, self.s_kW, self.s_M10 = S(), S(), S()   # taps kH * kW; kW; ceil(1024 / kW): r / kW = (r * M10) >> 10 for r < 49
            self.s_pok = [S(2), S(2)]                      # prologue: lanes whose first / second output pixel exists (ragged last tile)
            self.s_m = S(2)
            self.s_HW4, self.s_W4, self.s_Cin = S(), S(), S()
            self.s_sc = scr.sub(0, 8)                      # (the scheduler constants are dead before conv_setup loads the geometry)
        elif c.b_kcontig:
            self.WB = [[triple(False) for _ in range(c.NPB)] for _ in range(2)]   # like A: [MFMA half][piece][stage]
        else:
            self.WB = [[triple(False) for _ in range(4)] for _ in range(c.NPB // 2)]   # [pair][element][stage]
        self.v_oob = V()            # 0x80000000: a buffer offset the bounds check always rejects (reads as 0)
        self.s_tm = S(2)            # lanes whose 16-byte piece of a k-contiguous operand is real data in the LAST K-tile
        self.s_ktail = S()
        self.s_em = [S(2) for _ in range(4)]   # K % 4 != 0: lanes whose element j of their piece is real data in the last K-tile
        if c.conv:     # (the convolution kernels are out of SGPRs: the tap state is dead by the epilogue)
            self.srdBias, self.s_epi = self.s_scr.sub(0, 4), self.s_scr.sub(4, 4)
        else:
            self.srdBias = S(4)                                      # fused epilogue: the bias view (base, -, bytes, flags)
            self.s_epi = S(4, align=4)                               # rowStrideBias, colStrideBias (elements), activation, -
            # batch strides in bytes (grid y = batch index): read and consumed in once(), long before the fused epilogue loads its fields
            self.s_bsA, self.s_bsBC = self.srdBias.sub(0, 2), self.s_epi
        if c.pipe:
            # pipelined tile transitions: bit 0 = this launch may pipeline (beta == 0, plain epilogue, K a multiple of BK, >= 3 K-tiles),
            # bit 1 = armed: the descriptors already point at the NEXT tile and (vC, srdCd) address the tile being finished
            self.s_pipe = S()
            self.srdCd = S(4)        # C from this wave's first row of the tile being finished: rows move on by the stores' scalar offset
        self.vVA = [V() for _ in range(c.NPA)]
        self.vVB = [V() for _ in range(c.NPB)] if not c.conv else []
        self.vC = [V() for _ in range(c.TN)]
        if c.debug:
            self.srdD = S(4)
            self.s_dslot = S()
            self.v_dbg = V()
        self.ndump = 0
        self.dump_names = []
        self.vT = [V(16)]
        blk = V(12, align=4)
        self.vt = [blk[i] for i in range(10)]
        # filler experiments (timing only): dummy data / address registers that alias temporaries the loop does not use
        self.vF, self.vFaddr, self.vFoff = blk.sub(4, 4), blk[10], blk[11]

    def alloc_sched(self):
        """scheduler state.  A persistent workgroup owns units [u0, u1) of the launch's unit sequence (unit = one K slice of one tile,
        tile-major): u0 = t0 * P + p0, u1 = t1 * P + pe.  It runs, in this order: the FIRST slices [0, pe) of tile t1 (their running sum
        is sent on to the next workgroup), its whole tiles, and last the END piece [p0, P) of tile t0, which continues the running sum
        received from the previous workgroup -- by then long since sent (sched_next)."""
        c, S = self.c, self.p.salloc
        self.s_sc = None
        if c.persistent:
            self.s_sc = S(8, align=4)                      # scheduler constants, (re)loaded where they are used
            self.s_vid, self.s_t0, self.s_p0, self.s_t1, self.s_pe, self.s_tcur, self.s_phase = (S() for _ in range(7))
            self.s_kb, self.s_Keff, self.s_mode, self.s_tile = S(), S(), S(), S()
            self.s_end, self.s_fin, self.s_pz = S(), S(), S()
        else:
            self.s_Keff = self.s_K
            self.s_tile = None
            if not c.conv:
                self.s_sc = S(8, align=4)
            if c.cpers:
                self.s_tcur, self.s_img = S(), S()         # the next unit of this workgroup; the image of the tile being loaded

    # ------------------------------------------------------------------ queue models -> counted waits
    def vm_issue(self, tag):
        self.vmq.append(tag)

    def vm_wait(self, tag):
        if tag not in self.vmq:
            return
        idx = self.vmq.index(tag)
        n = len(self.vmq) - 1 - idx
        assert n <= 63
        if "vmwaits" not in self.c.ablate:
            self.p.emit("s_waitcnt", vmcnt=n)
        del self.vmq[:idx + 1]

    def lg_issue(self, tag):
        self.lgq.append(tag)

    def lg_wait(self, tags=None):
        """wait until every outstanding LDS op whose tag is in `tags` (None: all) has completed"""
        idxs = [i for i, t in enumerate(self.lgq) if tags is None or t in tags]
        if not idxs:
            return
        idx = max(idxs)
        n = min(len(self.lgq) - 1 - idx, 15)   # (the counter saturates the 4-bit field: waiting for more is still correct)
        self.p.emit("s_waitcnt", lgkmcnt=n)
        del self.lgq[:len(self.lgq) - n]

    # ------------------------------------------------------------------ small helpers
    @staticmethod
    def lds_at(entry):
        """(address register, extra byte offset) of one stage's entry of an address triple (Cfg.il: the stage is an immediate offset)"""
        return entry if isinstance(entry, tuple) else (entry, 0)

    def kq_swz(self, dst, x, tmp):
        """dst = kq_swz<BK>(x): BK 32: ((x>>1) ^ (x&1)) & 7; BK 16: ((x>>2) ^ ((x>>1)&1)) & 3"""
        e = self.p.emit
        if self.c.BK == 32:
            e("v_lshrrev_b32", dst, 1, x)
            e("v_and_b32", tmp, 1, x)
            e("v_xor_b32", dst, dst, tmp)
            e("v_and_b32", dst, 7, dst)
        else:
            e("v_lshrrev_b32", dst, 2, x)
            e("v_bfe_u32", tmp, x, 1, 1)
            e("v_xor_b32", dst, dst, tmp)
            e("v_and_b32", dst, 3, dst)

    def kq_row(self, dst, x, tmp):
        e = self.p.emit
        e("v_bfe_u32", tmp, x, 2, 1)
        e("v_xor_b32", dst, x, tmp)

    def dump(self, name, reg):
        """debug builds: workgroup 0 stores `reg` of every lane to slot `ndump` of the debug buffer (slot = 256 dwords)"""
        if not self.c.debug:
            return
        e = self.p.emit
        skip = self.p.label("nodump")
        e("s_cmp_lg_u32", s(2), 0)
        e("s_cbranch_scc1", skip)
        src = reg
        if isinstance(reg, Reg) and reg.kind == "s":
            e("v_mov_b32", self.vt[9], reg)
            src = self.vt[9]
        elif isinstance(reg, Reg) and reg.kind == "a":
            e("v_accvgpr_read_b32", self.vt[9], reg)
            src = self.vt[9]
        e("s_mov_b32", self.s_dslot, self.ndump * 1024)
        e("buffer_store_dword", src, self.v_dbg, self.srdD, self.s_dslot, offen=True)
        self.p.place(skip)
        self.dump_names.append(name)
        self.ndump += 1

    def dump_lds(self, name, byte_off):
        """debug: every lane reads LDS dword byte_off + 4*tid and dumps it"""
        if not self.c.debug:
            return
        e = self.p.emit
        e("v_add_u32", self.vt[8], byte_off, self.v_dbg)
        e("ds_read_b32", self.vt[8], self.vt[8])
        e("s_waitcnt", lgkmcnt=0)
        self.dump(name, self.vt[8])

    # ------------------------------------------------------------------ workgroup -> tile: arithmetic, no table
    def udiv(self, dst, x, magic):
        """dst = x / d for the divisor whose magic number (magic_u32) is in SGPR `magic`; dst != x"""
        e = self.p.emit
        e("s_mul_hi_u32", dst, x, magic)
        e("s_cmp_eq_u32", magic, 0)
        e("s_cselect_b32", dst, x, dst)

    def xcd_remap(self, dst, wg, xq, xr, tmp):
        """hardware sends workgroup g to XCD g % 8: give every XCD one contiguous chunk of ids (bijective for any grid size), so
        that the workgroups resident on an XCD work on neighbouring tiles and share operand panels in that XCD's L2.
        dst = (g % 8) * xq + min(g % 8, xr) + g / 8 with xq = G / 8, xr = G % 8; xq == 0: dst = g"""
        e = self.p.emit
        e("s_and_b32", tmp, wg, 7)
        e("s_mul_i32", dst, tmp, xq)
        e("s_min_u32", tmp, tmp, xr)
        e("s_add_u32", dst, dst, tmp)
        e("s_lshr_b32", tmp, wg, 3)
        e("s_add_u32", dst, dst, tmp)
        e("s_cmp_eq_u32", xq, 0)
        e("s_cselect_b32", dst, wg, dst)

    def tile_coords(self, tile, sc, pm, pn, tmp):
        """grouped raster (the ic / jr partition of gemm.nim:160-176 as arithmetic): tiles are walked in groups of group_m tile rows,
        column by column inside a group.  sc = the 8 SGPRs loaded from KA_SCHED; pm, pn, tmp[0..2]: distinct SGPRs, not `tile`"""
        e = self.p.emit
        tiles_m, tiles_n, gm, gsz_last, mg_width, mg_gm, mg_last = (sc[i] for i in range(7))
        e("s_mul_i32", tmp[0], gm, tiles_n)                   # width = tiles of a full group
        self.udiv(tmp[1], tile, mg_width)                     # group
        e("s_mul_i32", tmp[0], tmp[1], tmp[0])
        e("s_sub_u32", tmp[0], tile, tmp[0])                  # rem = index inside the group
        e("s_mul_i32", tmp[1], tmp[1], gm)                    # first tile row of the group
        e("s_sub_u32", tmp[2], tiles_m, tmp[1])
        e("s_cmp_lt_u32", tmp[2], gm)                         # the last group may have fewer rows
        e("s_cselect_b32", tmp[2], gsz_last, gm)              # gsz
        e("s_cselect_b32", pm, mg_last, mg_gm)                # its magic
        self.udiv(pn, tmp[0], pm)                             # pid_n = rem / gsz
        e("s_mul_i32", tmp[2], pn, tmp[2])
        e("s_sub_u32", tmp[0], tmp[0], tmp[2])                # rem % gsz
        e("s_add_u32", pm, tmp[1], tmp[0])                    # pid_m

    # ------------------------------------------------------------------ scheduler of the persistent kernels
    def unit_start(self, dst, vid, tmp):
        """first unit of workgroup `vid`: vid * units_q + floor(vid * units_r / workgroups) (s_sc holds the KA_SCHED2 block)"""
        e, sc = self.p.emit, self.s_sc
        e("s_mul_i32", tmp, vid, sc[4])
        self.udiv(dst, tmp, sc[7])
        e("s_mul_i32", tmp, vid, sc[3])
        e("s_add_u32", dst, dst, tmp)

    def sched_init(self):
        """[u0, u1) of this workgroup -> (t0, p0), (t1, pe)"""
        c, p = self.c, self.p
        e, st, sc = p.emit, self.s_t, self.s_sc
        L_two, L_have, L_str, L_end = p.label("twolevel"), p.label("haverange"), p.label("strided"), p.label("schedinit")
        e("s_load_dword", st[5], s(0, 2), KA_SCHED + 28)
        e("s_load_dwordx8", sc, s(0, 2), KA_SCHED2)
        e("s_waitcnt", lgkmcnt=0)
        P, mg_P = sc[1], sc[2]
        u0, u1 = st[2], st[3]
        e("s_bitcmp1_b32", sc[6], 2)
        e("s_cbranch_scc1", L_str)
        e("s_bitcmp1_b32", sc[6], 1)
        e("s_cbranch_scc1", L_two)
        # one level: virtual id (XCD remap) -> equal shares of all units
        self.xcd_remap(self.s_vid, s(2), st[5], sc[0], st[0])
        self.unit_start(u0, self.s_vid, st[0])
        e("s_add_u32", st[1], self.s_vid, 1)
        self.unit_start(u1, st[1], st[0])
        e("s_branch", L_have)
        # two levels: XCD x = id % 8 owns whole tiles; local index l = id / 8 shares that XCD's units (st[5] = workgroups / 8)
        p.place(L_two)
        Gq, T, mg = st[5], sc[3], sc[7]
        e("s_and_b32", st[0], s(2), 7)                      # x
        e("s_lshr_b32", st[1], s(2), 3)                     # l
        e("s_mul_i32", self.s_vid, st[0], Gq)
        e("s_add_u32", self.s_vid, self.s_vid, st[1])       # slot id: x * Gq + l
        e("s_mul_i32", st[4], T, st[0])
        e("s_lshr_b32", st[4], st[4], 3)                    # first tile of the XCD
        e("s_add_u32", st[0], st[0], 1)
        e("s_mul_i32", st[0], T, st[0])
        e("s_lshr_b32", st[0], st[0], 3)
        e("s_sub_u32", st[0], st[0], st[4])                 # its number of tiles
        e("s_mul_i32", st[0], st[0], P)                     # Ux = its units
        e("s_mul_i32", st[4], st[4], P)                     # its first unit
        self.udiv(self.s_t0, st[0], mg)                     # qx = Ux / Gq            (s_t0, s_t1, s_pe: free until set below)
        e("s_mul_i32", self.s_t1, self.s_t0, Gq)
        e("s_sub_u32", self.s_t1, st[0], self.s_t1)         # rx
        for dst, off in ((u0, 0), (u1, 1)):
            e("s_add_u32", st[0], st[1], off)               # l (+ 1)
            e("s_mul_i32", self.s_pe, st[0], self.s_t1)
            self.udiv(dst, self.s_pe, mg)                   # floor(l * rx / Gq)
            e("s_mul_i32", st[0], st[0], self.s_t0)
            e("s_add_u32", dst, dst, st[0])
            e("s_add_u32", dst, dst, st[4])
        p.place(L_have)
        self.udiv(self.s_t0, u0, mg_P)
        e("s_mul_i32", st[0], self.s_t0, P)
        e("s_sub_u32", self.s_p0, u0, st[0])
        self.udiv(self.s_t1, u1, mg_P)
        e("s_mul_i32", st[0], self.s_t1, P)
        e("s_sub_u32", self.s_pe, u1, st[0])
        e("s_cmp_lg_u32", self.s_p0, 0)
        e("s_cselect_b32", st[0], 1, 0)
        e("s_add_u32", self.s_tcur, self.s_t0, st[0])      # first whole tile
        e("s_cmp_eq_u32", u0, u1)
        e("s_cselect_b32", self.s_phase, 4, 0)             # (an empty range -- more workgroups than an XCD has units -- has nothing to do)
        e("s_branch", L_end)
        # strided whole tiles (flags bit 2; +44 = the stride = workgroups, +48 = tiles): workgroup v walks tiles v, v + G, v + 2G ... --
        # at any moment the chip works on G consecutive tiles of the raster, an XCD on one contiguous chunk of them, exactly like the
        # rounds of a one-tile-per-workgroup launch (contiguous ranges would put every workgroup of an XCD on panels of its own) --
        # and a workgroup goes from one tile to the next without leaving its K loop (Cfg.pipe)
        p.place(L_str)
        self.xcd_remap(self.s_vid, s(2), st[5], sc[0], st[0])
        e("s_mov_b32", self.s_t0, self.s_vid)
        e("s_mov_b32", self.s_tcur, self.s_vid)
        e("s_mov_b32", self.s_p0, 0)
        e("s_mov_b32", self.s_pe, 0)
        e("s_mov_b32", self.s_t1, sc[4])
        e("s_cmp_ge_u32", self.s_vid, sc[4])
        e("s_cselect_b32", self.s_phase, 4, 0)
        p.place(L_end)

    def tile_step(self, dst):
        """dst = distance between two whole tiles of this workgroup: 1, or the stride of a strided launch (s_sc holds KA_SCHED2)"""
        e, sc = self.p.emit, self.s_sc
        e("s_bitcmp1_b32", sc[6], 2)
        e("s_cselect_b32", dst, sc[3], 1)

    def sched_next(self, L_exit, L_recv):
        """the next run of this workgroup (falls through with s_tile, s_kb, s_Keff, s_mode, s_end set; L_exit when there is none).
        phase 0: slices [0, pe) of tile t1, running sum SENT to workgroup vid + 1 (slot vid) -- first, so that the receiver, which
                 needs it last, never waits;
        phase 1: whole tiles t0 (+1 if p0 > 0) .. t1 - 1;
        phase 2: the piece [p0, pz) of tile t0 (pz = P, or pe when the whole range lies inside one tile): continues the sum RECEIVED
                 from workgroup vid - 1.  Laser-order: if the sum has already arrived it becomes the running sum and the piece is one
                 run; else the piece's first slice is computed while the sender still works (run A: a chain from +0 needs nothing),
                 the sum is received after it, and the remaining slices follow as run B on top of it (mode CONT) -- every slice sum
                 is added in ascending order either way (gemm.nim:150-158).  One chain: one run, the received partial sum added last."""
        c, p = self.c, self.p
        e, st, sc, t = p.emit, self.s_t, self.s_sc, self.vt
        L_ph1, L_ph2, L_go, L_more = p.label("ph1"), p.label("ph2"), p.label("go"), p.label("whole")
        e("s_load_dwordx8", sc, s(0, 2), KA_SCHED2)
        e("s_waitcnt", lgkmcnt=0)
        P, slen = sc[1], sc[5]
        ke = st[0]
        # ---- phase 0: the first slices of the tile the range ends in
        e("s_cmp_lg_u32", self.s_phase, 0)
        e("s_cbranch_scc1", L_ph1)
        e("s_mov_b32", self.s_phase, 1)
        e("s_cmp_eq_u32", self.s_pe, 0)
        e("s_cbranch_scc1", L_ph1)
        e("s_cmp_lg_u32", self.s_t0, self.s_t1)
        e("s_cselect_b32", st[1], 1, 0)
        e("s_cmp_eq_u32", self.s_p0, 0)
        e("s_cselect_b32", st[2], 1, 0)
        e("s_or_b32", st[1], st[1], st[2])                  # t0 != t1 or p0 == 0: slice 0 of tile t1 is ours
        e("s_cmp_eq_u32", st[1], 0)
        e("s_cbranch_scc1", L_ph1)
        e("s_mov_b32", self.s_tile, self.s_t1)
        e("s_mov_b32", self.s_kb, 0)
        e("s_mul_i32", ke, self.s_pe, slen)
        e("s_mov_b32", self.s_mode, MODE_NORMAL)
        e("s_mov_b32", self.s_end, END_SEND)
        e("s_branch", L_go)
        # ---- phase 1: whole tiles
        p.place(L_ph1)
        e("s_cmp_lg_u32", self.s_phase, 1)
        e("s_cbranch_scc1", L_ph2)
        e("s_cmp_lt_u32", self.s_tcur, self.s_t1)
        e("s_cbranch_scc1", L_more)
        e("s_mov_b32", self.s_phase, 2)
        e("s_branch", L_ph2)
        p.place(L_more)
        e("s_mov_b32", self.s_tile, self.s_tcur)
        self.tile_step(st[1])
        e("s_add_u32", self.s_tcur, self.s_tcur, st[1])
        e("s_mov_b32", self.s_kb, 0)
        e("s_mov_b32", ke, self.s_K)
        e("s_mov_b32", self.s_mode, MODE_NORMAL)
        e("s_mov_b32", self.s_end, END_EPI)
        e("s_branch", L_go)
        # ---- phase 2: the piece that continues the previous workgroup's running sum
        p.place(L_ph2)
        e("s_cmp_lg_u32", self.s_phase, 2)
        e("s_cbranch_scc1", L_exit)
        e("s_mov_b32", self.s_phase, 4)
        e("s_cmp_eq_u32", self.s_p0, 0)
        e("s_cbranch_scc1", L_exit)
        e("s_mov_b32", self.s_tile, self.s_t0)
        e("s_cmp_eq_u32", self.s_t0, self.s_t1)
        e("s_cselect_b32", self.s_pz, self.s_pe, P)
        e("s_cmp_eq_u32", self.s_pz, P)
        e("s_cselect_b32", self.s_fin, END_EPI, END_SEND)
        e("s_mul_i32", self.s_kb, self.s_p0, slen)
        if c.exact:
            L_late = p.label("late")
            # has the sum arrived?  (one look at the flag; sc[6] bit 0: tests force the two-run path)
            e("s_bitcmp1_b32", sc[6], 0)
            e("s_cbranch_scc1", L_late)
            e("s_sub_u32", st[3], self.s_vid, 1)
            self.ws_descriptors(st[3])
            e("buffer_load_dword", t[9], OFF, self.srdA, 0, sc1=True)
            e("s_waitcnt", vmcnt=0)
            # the four waves must take the same path (each has looked at the flag at its own time): every wave posts what it saw,
            # the sum counts as arrived only if all four saw it
            e("s_lshl_b32", st[4], self.s_wave, 2)
            e("s_add_u32", st[4], st[4], c.VOTE)
            e("v_mov_b32", t[8], st[4])
            e("ds_write_b32", t[8], t[9])
            e("s_waitcnt", lgkmcnt=0)
            e("s_barrier")
            e("v_mov_b32", t[8], c.VOTE)
            e("ds_read_b128", v(t[4].idx, 4), t[8])
            e("s_waitcnt", lgkmcnt=0)
            e("v_min_u32", t[4], t[4], t[5])
            e("v_min_u32", t[6], t[6], t[7])
            e("v_min_u32", t[4], t[4], t[6])
            e("s_nop", 1)
            e("v_readfirstlane_b32", st[4], t[4])
            e("s_load_dwordx8", sc, s(0, 2), KA_SCHED2)
            e("s_waitcnt", lgkmcnt=0)
            e("s_cmp_eq_u32", st[4], 0)
            e("s_cbranch_scc1", L_late)
            e("s_mov_b32", self.s_phase, 5)                  # the receive block comes back to the run set-up
            e("s_mul_i32", ke, self.s_pz, slen)
            e("s_min_u32", ke, ke, self.s_K)
            e("s_sub_u32", self.s_Keff, ke, self.s_kb)
            e("s_mov_b32", self.s_mode, MODE_CONT)
            e("s_mov_b32", self.s_end, self.s_fin)
            e("s_branch", L_recv)
            p.place(L_late)
            e("s_add_u32", ke, self.s_p0, 1)
            e("s_mul_i32", ke, ke, slen)
            e("s_mov_b32", self.s_mode, MODE_ACC)
            e("s_mov_b32", self.s_end, END_RECV)
        else:
            e("s_mul_i32", ke, self.s_pz, slen)
            e("s_mov_b32", self.s_mode, MODE_NORMAL)
            e("s_mov_b32", self.s_end, END_RECV)
        p.place(L_go)
        e("s_min_u32", ke, ke, self.s_K)
        e("s_sub_u32", self.s_Keff, ke, self.s_kb)

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        """once per workgroup: kernel arguments, lane decode, LDS addresses; then (persistent kernels: per run) the tile, its
        descriptors and the first two K-tiles"""
        c, p = self.c, self.p
        e = p.emit
        st = self.s_t
        self.once()
        self.L_run, self.L_exit = p.label("next_run"), p.label("exit")
        self.L_setup, self.L_recv = p.label("setup"), p.label("recv")
        if c.persistent:
            self.sched_init()
            p.place(self.L_run)
            self.sched_next(self.L_exit, self.L_recv)
            p.place(self.L_setup)
            e("s_barrier", comment="every wave is done with the previous run's LDS tiles")
            # (KA_TAB's low word, GEMM kernels: the first tile of this launch -- a launch may cover the tiles [base, base + T') of the
            # raster only, the two launches of the launcher's hybrid plan; 0 otherwise.  The scheduler's tile numbers stay relative.)
            e("s_add_u32", st[4], self.s_tile, self.ka0[6])
            tile = st[4]
            e("s_load_dwordx8", self.s_sc, s(0, 2), KA_SCHED)
            e("s_waitcnt", lgkmcnt=0)
        elif c.cpers:
            e("s_mov_b32", self.s_tcur, s(2))
            p.place(self.L_run)
            e("s_barrier", comment="every wave is done with the previous unit's LDS tiles and tap table")
            self.next_unit(self.L_exit)
            self.run_setup()
            return
        else:
            e("s_load_dwordx8", self.s_sc, s(0, 2), KA_SCHED)
            e("s_load_dword", st[5], s(0, 2), KA_SCHED2)
            e("s_waitcnt", lgkmcnt=0)
            self.xcd_remap(st[4], s(2), self.s_sc[7], st[5], st[0])
            if not c.conv:
                e("s_add_u32", st[4], st[4], self.ka0[6])       # (the launch's first tile, see above)
            tile = st[4]
        self.tile_coords(tile, self.s_sc, st[0], st[1], (st[2], st[3], st[5]))
        self.dump("pid_m", st[0])
        self.dump("pid_n", st[1])
        e("s_mul_i32", self.s_m0, st[0], c.BM)
        e("s_mul_i32", self.s_n0, st[1], c.BN)
        self.run_setup()

    def once(self):
        c, p = self.c, self.p
        e = p.emit
        t = self.vt
        st = self.s_t
        p.note(f"{c.name}: {c.BM}x{c.BN}x{c.BK} tile, 4 waves (one per SIMD), wave tile {c.WTM}x{c.WTN}, "
               f"{'laser-order (kc = 512 slices)' if c.exact else 'one accumulation chain'}")
        e("s_load_dwordx8", self.ka0, s(0, 2), KA_A)
        e("s_load_dwordx8", self.ka1, s(0, 2), KA_LDA)
        if not c.conv:
            # batched problems (gemm_strided_batched; the kc slices of the slice-parallel form): workgroup id y = batch index,
            # operand b at base + b * batch stride (bytes, 64-bit; 0 for plain launches)
            e("s_load_dwordx2", self.s_bsA, s(0, 2), KA_BSA)
            e("s_load_dwordx4", self.s_bsBC, s(0, 2), KA_CONV1 + 8)
            e("s_waitcnt", lgkmcnt=0)
            for ptr, bs in ((self.ka0.sub(0, 2), self.s_bsA), (self.ka0.sub(2, 2), self.s_bsBC.sub(0, 2)), (self.ka0.sub(4, 2), self.s_bsBC.sub(2, 2))):
                e("s_mul_i32", st[2], s(3), bs[0])
                e("s_mul_hi_u32", st[3], s(3), bs[0])
                e("s_mul_i32", st[4], s(3), bs[1])
                e("s_add_u32", st[3], st[3], st[4])
                e("s_add_u32", ptr[0], ptr[0], st[2])
                e("s_addc_u32", ptr[1], ptr[1], st[3])
        e("s_waitcnt", lgkmcnt=0)
        if c.pipe:
            self.pipe_eligible()
        if c.pre:
            e("s_load_dword", st[2], s(0, 2), KA_PRE)
            e("s_waitcnt", lgkmcnt=0)
            for bit, m in ((0, self.s_preA), (1, self.s_preB)):
                e("s_bitcmp1_b32", st[2], bit)
                e("s_cselect_b32", m[0], -1, 0)
                e("s_cselect_b32", m[1], -1, 0)
        if c.debug:
            e("s_load_dwordx2", self.srdD.sub(0, 2), s(0, 2), KA_DBG)
            e("s_waitcnt", lgkmcnt=0)
            e("s_and_b32", self.srdD[1], self.srdD[1], 0xffff)
            e("s_mov_b32", self.srdD[2], 0x10000000)
            e("s_mov_b32", self.srdD[3], 0x00020000)
            e("v_lshlrev_b32", self.v_dbg, 2, v(0))
            self.dump("tid", v(0))
            self.dump("wgid_x", s(2))
            self.dump("lda", self.s_lda)
            self.dump("K", self.s_K)
        # ---- lane decode (independent of the tile) ----
        tid = v(0)
        lane, lo, hi, lor, kqs = t[0], t[1], t[2], t[3], t[4]
        e("v_and_b32", lane, 63, tid)
        e("v_lshrrev_b32", t[5], 6, tid)
        e("s_nop", 1, comment="VALU write -> v_readfirstlane of the same VGPR needs wait states (seen on hardware: it read the old value)")
        e("v_readfirstlane_b32", self.s_wave, t[5])
        e("s_nop", 3)
        e("v_and_b32", lo, 31, lane)
        e("v_lshrrev_b32", hi, 5, lane)
        self.kq_row(lor, lo, t[5])
        self.kq_swz(kqs, lo, t[5])
        e("s_lshr_b32", st[2], self.s_wave, 1)
        e("s_mul_i32", self.s_wm0, st[2], c.WTM)
        e("s_and_b32", st[2], self.s_wave, 1)
        e("s_mul_i32", self.s_wn0, st[2], c.WTN)
        # fragment reads of group g: (wm0 + lor) * BK * 4 [+ BK*BM*4 + (wn0 + lor) * BK * 4 for B] + 16 * ((2g + hi) ^ kqs)
        e("v_add_u32", t[5], self.s_wm0, lor)
        e("v_mul_u32_u24", t[6], c.ROWP, t[5])
        if c.LDS0:
            e("v_add_u32", t[6], c.LDS0, t[6])
        e("v_add_u32", t[5], self.s_wn0, lor)
        e("v_mul_u32_u24", t[7], c.ROWP, t[5])
        e("v_add_u32", t[7], c.LDS0 + c.ROWP * c.BM, t[7])
        for g in range(c.NG):
            e("v_or_b32", t[5], 2 * g, hi)
            e("v_xor_b32", t[5], t[5], kqs)
            e("v_lshlrev_b32", t[5], 4, t[5])
            for R, row in ((self.RA, t[6]), (self.RB, t[7])):
                if c.il:
                    e("v_add_u32", R[g][0][0], t[5], row)
                    continue
                e("v_add_u32", R[g][0], t[5], row)
                e("v_add_u32", R[g][1], c.STAGE, R[g][0])
                e("v_add_u32", R[g][2], 2 * c.STAGE, R[g][0])
        # k-contiguous operand (A always; B when it is passed transposed): a 16-byte piece = 4 consecutive k of row x,
        # kq = tid % (BK/4), x = tid / (BK/4) (+ XS per piece)
        nkq = c.BK // 4
        kq, x, row, sw = t[0], t[1], t[2], t[3]
        e("v_and_b32", kq, nkq - 1, tid)
        e("v_lshrrev_b32", x, nkq.bit_length() - 1, tid)
        self.kq_row(row, x, t[5])
        self.kq_swz(sw, x, t[5])
        e("v_mov_b32", self.v_oob, 0x80000000)

        def kcontig_lds(W, NP, lds_off):
            e("v_lshrrev_b32", t[5], 1, kq)          # kq / 2
            e("v_lshlrev_b32", t[5], 1, t[5])        # 2 * (kq / 2)
            e("v_and_b32", t[6], 1, kq)              # kq % 2
            e("v_mul_u32_u24", t[7], c.ROWP, row)    # row * (bytes between rows)
            e("v_lshl_add_u32", t[7], t[6], 3, t[7])  # + 8 * (kq % 2)
            if lds_off:
                e("v_add_u32", t[7], lds_off, t[7])
            XSB = (256 // nkq) * c.ROWP              # bytes between two pieces of a thread: XS rows
            for cc in range(2):
                e("v_or_b32", t[8], cc, t[5])
                e("v_xor_b32", t[8], t[8], sw)
                w0 = self.lds_at(W[cc][0][2])[0]
                e("v_lshl_add_u32", w0, t[8], 4, t[7])
                for pi in range(NP):
                    if pi:
                        e("v_add_u32", self.lds_at(W[cc][pi][2])[0], XSB * pi, w0)
                    if c.il:
                        continue
                    e("v_add_u32", W[cc][pi][0], c.STAGE, W[cc][pi][2])
                    e("v_add_u32", W[cc][pi][1], 2 * c.STAGE, W[cc][pi][2])

        kcontig_lds(self.WA, c.NPA, c.LDS0)
        e("s_lshl_b32", st[5], self.s_ldb, 2, comment="ldb * 4 bytes")
        if c.b_kcontig:
            kcontig_lds(self.WB, c.NPB, c.ROWP * c.BM)
            e("s_mov_b32", self.s_bstep, c.BK * 4, comment="B (stored transposed) advances BK elements along its rows per K-tile")
        if not c.b_kcontig and not c.conv:
            # B pieces (x-contiguous, 16 B = 4 consecutive x of row k), handled in pairs (k, k+2) -- DESIGN.md 3.2 pair mode
            aa, pp, hh, c0, xq, kb0 = t[0], t[1], t[2], t[3], t[4], t[6]
            BX16 = c.BN // 16
            KG = 8 * 16 // BX16
            e("v_and_b32", aa, 3, tid)
            e("v_bfe_u32", pp, tid, 2, 1)
            e("v_bfe_u32", hh, tid, 3, 1)
            e("v_lshrrev_b32", c0, 4, tid)
            e("v_and_b32", t[5], BX16 - 1, c0)
            e("v_lshl_add_u32", xq, t[5], 2, aa)                 # xq = (c0 % BX16) * 4 + a
            e("v_lshrrev_b32", t[5], BX16.bit_length() - 1, c0)  # c0 / BX16
            e("v_lshlrev_b32", t[5], 3, t[5])
            e("v_lshl_add_u32", kb0, hh, 2, t[5])
            e("v_add_u32", kb0, kb0, pp)                         # kb0 = 8 * (c0 / BX16) + 4h + p
            e("v_mul_lo_u32", t[7], kb0, st[5])
            e("v_lshl_add_u32", self.vVB[0], xq, 4, t[7])        # VB(gi = 0, j = 0)
            e("s_lshl_b32", st[4], st[5], 1)                     # 2 * ldb * 4
            e("s_mul_i32", st[3], st[5], KG)                     # KG * ldb * 4
            for gi in range(c.NPB // 2):
                if gi:
                    e("v_add_u32", self.vVB[2 * gi], st[3], self.vVB[2 * gi - 2])
                e("v_add_u32", self.vVB[2 * gi + 1], st[4], self.vVB[2 * gi])
            e("s_mul_i32", self.s_bstep, st[5], c.BK, comment="B advances BK rows per K-tile")
            # LDS write addresses of the B pairs: for element e of the pieces: x = 4xq + e, row = 4xq + (e ^ (xq & 1)),
            #   L = 2 * (k / 8) + (k & 1), word = (k % 8) >> 1;  WB = BK*BM*4 + row*BK*4 + 16 * (L ^ kq_swz(x)) + 4 * word
            e("s_mov_b32", st[0], c.ROWP * c.BM, comment="the B panel follows the A panel in a stage")
            for gi in range(c.NPB // 2):
                kk = t[7]
                e("v_add_u32", kk, gi * KG, kb0)
                e("v_lshrrev_b32", t[8], 3, kk)
                e("v_lshlrev_b32", t[8], 1, t[8])
                e("v_and_b32", t[9], 1, kk)
                e("v_or_b32", t[8], t[8], t[9])                  # L
                e("v_bfe_u32", t[9], kk, 1, 2)                   # word = (k & 7) >> 1  (k % 8 in {0,1,4,5} -> 0 or 2)
                e("v_lshlrev_b32", t[9], 2, t[9])                # 4 * word
                for ee in range(4):
                    xx, rr, ss = t[0], t[1], t[2]  # aa / pp / hh are dead from here on
                    e("v_lshl_add_u32", xx, xq, 2, ee)           # x = 4xq + e
                    self.kq_row(rr, xx, t[5])
                    self.kq_swz(ss, xx, t[5])
                    e("v_xor_b32", ss, ss, t[8])                 # L ^ swz
                    e("v_mul_u32_u24", rr, c.ROWP, rr)
                    e("v_lshl_add_u32", rr, ss, 4, rr)
                    e("v_add3_u32", self.lds_at(self.WB[gi][ee][2])[0], rr, t[9], st[0])
                    if c.il:
                        continue
                    e("v_add_u32", self.WB[gi][ee][0], c.STAGE, self.WB[gi][ee][2])
                    e("v_add_u32", self.WB[gi][ee][1], 2 * c.STAGE, self.WB[gi][ee][2])

    def pipe_eligible(self):
        """s_pipe bit 0: tile transitions of this launch may be pipelined -- beta == 0 (the running sum of the next tile starts at 0:
        nothing to fetch), no bias / activation (the plain store), K a multiple of BK (no K-tail masks to undo between tiles) and at
        least three K-tiles (the switch happens two tile bodies before the end of a tile)"""
        c, e, st = self.c, self.p.emit, self.s_t
        e("s_load_dwordx2", self.srdBias.sub(0, 2), s(0, 2), KA_BIAS)
        e("s_load_dword", st[2], s(0, 2), KA_EPI + 8)
        e("s_waitcnt", lgkmcnt=0)
        e("s_or_b32", st[3], self.srdBias[0], self.srdBias[1])
        e("s_or_b32", st[3], st[3], st[2])
        e("s_and_b32", st[4], self.s_beta, 0x7fffffff)
        e("s_or_b32", st[3], st[3], st[4])
        e("s_and_b32", st[4], self.s_K, c.BK - 1)
        e("s_or_b32", st[3], st[3], st[4])
        e("s_cmp_eq_u32", st[3], 0)
        e("s_cselect_b32", self.s_pipe, 1, 0)
        e("s_cmp_lt_u32", self.s_K, 3 * c.BK)
        e("s_cselect_b32", self.s_pipe, 0, self.s_pipe)

    def kcontig_goff(self, Voff, NP, ld_bytes):
        """global offsets of the 16-byte pieces of a k-contiguous operand: V_i = (x + i*XS) * ld * 4 + kq * 16 (per run: the K tail
        of a run overwrites them with the out-of-bounds offset)"""
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        nkq = c.BK // 4
        XS = 256 // nkq
        e("v_and_b32", t[0], nkq - 1, v(0))
        e("v_lshrrev_b32", t[1], nkq.bit_length() - 1, v(0))
        e("v_mul_lo_u32", t[7], t[1], ld_bytes)
        e("v_lshl_add_u32", Voff[0], t[0], 4, t[7])
        e("s_mul_i32", st[4], ld_bytes, XS)
        for i in range(1, NP):
            e("v_add_u32", Voff[i], st[4], Voff[i - 1])

    def ab_descriptors(self, again=False):
        """srdA / srdB of the run k in [kb, kb + Keff) of tile (m0, n0).  Clobbers s_t[0], [2], [3], [5]."""
        c, p = self.c, self.p
        e = p.emit
        st = self.s_t
        Keff = self.s_Keff
        e("s_lshl_b32", st[3], self.s_lda, 2, comment="lda * 4 bytes")
        e("s_lshl_b32", st[5], self.s_ldb, 2, comment="ldb * 4 bytes")
        A_, B_, C_ = self.ka0.sub(0, 2), self.ka0.sub(2, 2), self.ka0.sub(4, 2)
        # A panel: base = A + m0 * lda * 4 + kb * 4; bytes = (min(M - m0, BM) - 1) * lda * 4 + Keff * 4
        e("s_mul_hi_u32", st[2], self.s_m0, st[3])
        e("s_mul_i32", st[0], self.s_m0, st[3])
        e("s_add_u32", self.srdA[0], A_[0], st[0])
        e("s_addc_u32", self.srdA[1], A_[1], st[2])
        if c.persistent:
            e("s_lshl_b32", st[0], self.s_kb, 2)
            e("s_add_u32", self.srdA[0], self.srdA[0], st[0])
            e("s_addc_u32", self.srdA[1], self.srdA[1], 0)
        e("s_and_b32", self.srdA[1], self.srdA[1], 0xffff)
        e("s_sub_u32", st[0], self.s_M, self.s_m0)
        e("s_min_u32", st[0], st[0], c.BM)
        e("s_sub_u32", st[0], st[0], 1)
        e("s_mul_i32", st[0], st[0], st[3])
        e("s_lshl_b32", st[2], Keff, 2)
        e("s_add_u32", self.srdA[2], st[0], st[2])
        e("s_mov_b32", self.srdA[3], 0x00020000)
        if c.conv:
            self.conv_setup(again)
            if c.debug:
                for r_ in range(0, 20, 4):
                    self.dump_lds(f"tab[{r_ * 256}+4tid]", r_ * 256)
                self.dump("conv k0", self.s_k0)
                self.dump("conv HW4", self.s_HW4)
                self.dump("conv Cin", self.s_Cin)
                self.dump("conv WB00", self.lds_at(self.WB[0][0][2])[0])
                self.dump("conv WB31", self.lds_at(self.WB[3][1][2])[0])
        elif c.b_kcontig:
            # B^T panel: base = B + n0 * ldb * 4 + kb * 4; bytes = (min(N - n0, BN) - 1) * ldb * 4 + Keff * 4
            e("s_mul_hi_u32", st[2], self.s_n0, st[5])
            e("s_mul_i32", st[0], self.s_n0, st[5])
            e("s_add_u32", self.srdB[0], B_[0], st[0])
            e("s_addc_u32", self.srdB[1], B_[1], st[2])
            if c.persistent:
                e("s_lshl_b32", st[0], self.s_kb, 2)
                e("s_add_u32", self.srdB[0], self.srdB[0], st[0])
                e("s_addc_u32", self.srdB[1], self.srdB[1], 0)
            e("s_and_b32", self.srdB[1], self.srdB[1], 0xffff)
            e("s_sub_u32", st[0], self.s_N, self.s_n0)
            e("s_min_u32", st[0], st[0], c.BN)
            e("s_sub_u32", st[0], st[0], 1)
            e("s_mul_i32", st[0], st[0], st[5])
            e("s_lshl_b32", st[2], Keff, 2)
            e("s_add_u32", self.srdB[2], st[0], st[2])
        else:
            # B panel: base = B + n0 * 4 + kb * ldb * 4; bytes = (Keff - 1) * ldb * 4 + (N - n0) * 4
            e("s_lshl_b32", st[0], self.s_n0, 2)
            e("s_add_u32", self.srdB[0], B_[0], st[0])
            e("s_addc_u32", self.srdB[1], B_[1], 0)
            if c.persistent:
                e("s_mul_hi_u32", st[2], self.s_kb, st[5])
                e("s_mul_i32", st[0], self.s_kb, st[5])
                e("s_add_u32", self.srdB[0], self.srdB[0], st[0])
                e("s_addc_u32", self.srdB[1], self.srdB[1], st[2])
            e("s_and_b32", self.srdB[1], self.srdB[1], 0xffff)
            e("s_sub_u32", st[0], Keff, 1)
            e("s_mul_i32", st[0], st[0], st[5])
            e("s_sub_u32", st[2], self.s_N, self.s_n0)
            e("s_lshl_b32", st[2], st[2], 2)
            e("s_add_u32", self.srdB[2], st[0], st[2])
        e("s_mov_b32", self.srdB[3], 0x00020000)

    def run_setup(self):
        """one run = k in [kb, kb + Keff) of tile (m0, n0): K-tail masks, piece offsets, descriptors, the first two K-tiles"""
        c, p = self.c, self.p
        e = p.emit
        t = self.vt
        st = self.s_t
        Keff = self.s_Keff
        # K tail: pieces of the last K-tile that lie beyond K get an offset the bounds check rejects (they read as 0, like
        # Laser's zero-padded panels, gemm_packing.nim:46-55)
        nkq = c.BK // 4
        e("s_and_b32", self.s_ktail, Keff, c.BK - 1)
        e("v_and_b32", t[5], nkq - 1, v(0))
        e("v_lshlrev_b32", t[5], 2, t[5])
        e("v_cmp_gt_u32", self.s_tm, self.s_ktail, t[5])
        # K % 4 != 0: the piece that straddles K is loaded whole (its tail belongs to the next row, or reads 0 past the panel) and
        # its elements beyond K are zeroed in the staging registers before they are stored (mask_last_pieces)
        for j in range(4):
            e("v_add_u32", t[6], j, t[5])
            e("v_cmp_gt_u32", self.s_em[j], self.s_ktail, t[6])
        e("s_lshl_b32", st[3], self.s_lda, 2, comment="lda * 4 bytes")
        self.kcontig_goff(self.vVA, c.NPA, st[3])
        e("s_lshl_b32", st[5], self.s_ldb, 2, comment="ldb * 4 bytes")
        if c.b_kcontig:
            self.kcontig_goff(self.vVB, c.NPB, st[5])
        e("s_nop", 4)
        self.ab_descriptors()
        # C: the whole matrix (conv: this image's [M][oH*oW] block), bytes = (M - 1) * ldc * 4 + N * 4
        self.c_descriptor()
        # number of K-tiles
        e("s_add_u32", self.s_rem, Keff, c.BK - 1)
        e("s_lshr_b32", self.s_rem, self.s_rem, (c.BK).bit_length() - 1)
        # ---- tile 0 -> LDS stage 0, tile 1 -> staging registers ----
        if c.debug:
            for k_ in range(4):
                self.dump(f"srdA[{k_}]", self.srdA[k_])
            for k_ in range(4):
                self.dump(f"srdB[{k_}]", self.srdB[k_])
            self.dump("vVA0", self.vVA[0])
            self.dump("vVA1", self.vVA[1])
            if not c.conv:
                self.dump("vVB0", self.vVB[0])
                self.dump("vVB1", self.vVB[1])
            self.dump("WA0", self.lds_at(self.WA[0][0][2])[0])
            self.dump("WA1", self.lds_at(self.WA[1][0][2])[0])
            self.dump("WB00", self.lds_at(self.WB[0][0][2])[0])
            self.dump("RA0", self.lds_at(self.RA[0][0])[0])
            self.dump("RB0", self.lds_at(self.RB[0][0])[0])
        if c.deep:
            self.first_tiles_deep()
            self.first_tiles_done()
            return
        L_slow, L_join = p.label("fewtiles"), p.label("tiles01")
        fast = not c.debug      # (round 6: the convolution kernels too -- their prologue used to pay two memory latencies per tile)
        if fast:
            # Three or more K-tiles (no tile of the first two is the ragged last one): tiles 0 AND 1 are requested back to back --
            # tile 0 into the fragment registers (idle until the first fragment read), tile 1 into the staging registers where the loop
            # expects it -- so a run starts after ONE memory latency instead of two (the request for tile 1 used to wait until
            # tile 0 had arrived and been stored).  The counted waits below leave tile 1's loads in flight.
            state = (list(self.vmq), list(self.lgq))
            e("s_cmp_lt_u32", self.s_rem, 3)
            e("s_cbranch_scc1", L_slow)
            pool = [r for slot in range(2) for r in (self.fa[slot] + self.fb[slot])]
            real = (self.stA, self.stB)
            if c.conv:      # (B's pieces are register PAIRS there: two to a fragment register quad)
                assert 4 * len(pool) >= 4 * c.NPA + 2 * c.NPB
                tmp = (pool[:c.NPA], [pool[c.NPA + i // 2].sub(2 * (i % 2), 2) for i in range(c.NPB)])
            else:
                assert len(pool) >= c.NPA + c.NPB
                tmp = (pool[:c.NPA], pool[c.NPA:c.NPA + c.NPB])
            self.stA, self.stB = tmp
            self.issue_loads_all()
            self.advance_srds()
            self.stA, self.stB = real
            self.issue_loads_all()
            self.advance_srds()
            self.stA, self.stB = tmp
            if c.conv:
                # (the gathers of a tile are requested before its filter pieces: stored in that order, so that every counted wait names
                # tile 0's loads; the table reads of tile 1's gathers were waited for before these stores were issued, in the other path
                # after them: all LDS operations drained here keeps the two paths' queues alike)
                self.run_ops([o for grp in self.conv_store_ops(2) for o in grp])
            for pi in range(c.NPA):
                self.store_A_piece(pi, k=2)
            if c.conv:
                self.lg_wait(None)
            elif c.b_kcontig:
                for pj in range(c.NPB):
                    self.store_B_kpiece(pj, k=2)
            else:
                for gi in range(c.NPB // 2):
                    self.store_B_pair(gi, k=2)
            self.stA, self.stB = real
            e("s_branch", L_join)
            fast_state = (list(self.vmq), list(self.lgq))
            self.vmq, self.lgq = state
            p.place(L_slow)
        self.tail_mask_if(self.s_rem, 1)        # a single K-tile: tile 0 is the last one
        self.issue_loads_all()
        self.mask_last_pieces_if(self.s_rem, 1)
        self.advance_srds()
        if c.debug:
            e("s_waitcnt", vmcnt=0)
            self.vmq.clear()
            for k_ in range(4):
                self.dump(f"stA0[{k_}]", self.stA[0][k_])
            for k_ in range(2 if c.conv else 4):
                self.dump(f"stB0[{k_}]", self.stB[0][k_])
            if c.conv:
                for i_ in range(8):
                    self.dump(f"conv koff[{i_}]", self.s_koff[i_])
                    self.dump(f"conv vB0[{i_}]", self.vB0[i_])
                    self.dump(f"conv vB1[{i_}]", self.vB1[i_])
                    self.dump(f"conv stB[{i_}][0]", self.stB[i_][0])
                    self.dump(f"conv stB[{i_}][1]", self.stB[i_][1])
            self.dump("stA_last[3]", self.stA[-1][3])
        for pi in range(c.NPA):
            self.store_A_piece(pi, k=2)     # tile 0 goes to LDS stage 0 = the "third" stage of the write triples
        if c.conv:
            self.run_ops([o for grp in self.conv_store_ops(2) for o in grp])
        elif c.b_kcontig:
            for pj in range(c.NPB):
                self.store_B_kpiece(pj, k=2)
        else:
            for gi in range(c.NPB // 2):
                self.store_B_pair(gi, k=2)
        if c.debug:
            self.lg_wait(None)
            e("s_barrier")
            for k_ in range(0, 8):
                self.dump_lds(f"lds[{k_ * 1024}+4tid]", c.LDS0 + k_ * 1024)
            self.dump_lds("ldsB[0+4tid]", c.LDS0 + c.BK * c.BM * 4)
            e("s_barrier")
        self.tail_mask_if(self.s_rem, 2)
        self.issue_loads_all()
        self.mask_last_pieces_if(self.s_rem, 2)
        self.advance_srds()
        if fast:
            assert (self.vmq, self.lgq) == fast_state, "the two prologue paths must leave the same loads and stores in flight"
            p.place(L_join)
        self.tail_mask_if(self.s_rem, 3)        # the first loop body loads tile 2
        self.first_tiles_done()

    def store_tile_to_lds(self, k):
        """every piece of the tile in the current staging registers -> LDS (write-address triple index k)"""
        c = self.c
        for pi in range(c.NPA):
            self.store_A_piece(pi, k=k)
        if c.b_kcontig:
            for pj in range(c.NPB):
                self.store_B_kpiece(pj, k=k)
        else:
            for gi in range(c.NPB // 2):
                self.store_B_pair(gi, k=k)

    def first_tiles_deep(self):
        """deep staging: tile 0 -> LDS stage 0, tile 1 -> register set 1, tile 2 -> register set 0, both still in flight at the loop's
        entry.  Four or more K-tiles: all three requested back to back (tile 0 into the fragment registers), one memory latency
        per run; fewer: one after the other, each with the masks of a ragged last tile."""
        c, p = self.c, self.p
        e = p.emit
        set0, set1 = self.st_sets
        L_slow, L_join = p.label("fewtiles"), p.label("tiles012")
        state = (list(self.vmq), list(self.lgq))
        e("s_cmp_lt_u32", self.s_rem, 4)
        e("s_cbranch_scc1", L_slow)
        pool = [r for slot in range(2) for r in (self.fa[slot] + self.fb[slot])]
        assert len(pool) >= c.NPA + c.NPB
        tmp = (pool[:c.NPA], pool[c.NPA:c.NPA + c.NPB])
        for regs in (tmp, set1, set0):
            self.stA, self.stB = regs
            self.issue_loads_all()
            self.advance_srds()
        self.stA, self.stB = tmp
        self.store_tile_to_lds(2)               # tile 0 goes to LDS stage 0 = the "third" stage of the write triples
        self.stA, self.stB = set0
        e("s_branch", L_join)
        fast_state = (list(self.vmq), list(self.lgq))
        self.vmq, self.lgq = state
        p.place(L_slow)
        for n, regs in ((1, set0), (2, set1), (3, set0)):
            self.stA, self.stB = regs
            self.tail_mask_if(self.s_rem, n)        # n K-tiles: tile n - 1 is the last one
            self.issue_loads_all()
            self.mask_last_pieces_if(self.s_rem, n)
            self.advance_srds()
            if n == 1:
                self.store_tile_to_lds(2)
        self.stA, self.stB = set0
        assert (self.vmq, self.lgq) == fast_state, "the two prologue paths must leave the same loads and stores in flight"
        p.place(L_join)
        self.tail_mask_if(self.s_rem, 4)        # the first loop body loads tile 3

    def first_tiles_done(self):
        c, p = self.c, self.p
        e = p.emit
        self.init_accumulators()
        self.lg_wait(None)
        e("s_barrier")
        self.read_group(0, 0, 0)
        if c.debug:
            self.lg_wait(None)
            for k_ in range(4):
                self.dump(f"fa[0][0][{k_}]", self.fa[0][0][k_])
            for k_ in range(4):
                self.dump(f"fb[0][0][{k_}]", self.fb[0][0][k_])
            for _ in range(c.TM + c.TN):
                self.lg_issue(("R", 0))   # (keeps the loop-carried queue model: the body's first wait becomes a no-op wait)
        if c.filler:
            e("v_lshlrev_b32", self.vFaddr, 4, v(0))
            e("v_add_u32", self.vFaddr, c.lds_bytes, self.vFaddr, comment="filler experiments: 16 B per lane past the kernel's LDS")
            e("v_lshlrev_b32", self.vFoff, 4, v(0))

    def init_accumulators(self):
        """the chain starts at +0; laser-order: the running sum starts as beta * C0 -- unless this run continues a received sum
        (MODE_CONT) or does not use it (MODE_ACC: a piece's first slice, computed before the sum is received)"""
        c, p = self.c, self.p
        e = p.emit
        for b in range(c.NB):
            for r in range(c.ACCR):
                e("v_accvgpr_write_b32", self.acc[b][r], 0)
        if not c.exact:
            return
        keep = p.label("keeprun")
        if c.persistent:
            e("s_cmp_lg_u32", self.s_mode, MODE_NORMAL)
            e("s_cbranch_scc1", keep)
        for b in range(c.NB):
            for r in range(c.ACCR):
                e("v_mov_b32" if c.runv else "v_accvgpr_write_b32", self.run[b][r], 0)
        self.load_beta_c()
        p.place(keep)

    def c_descriptor(self):
        """srdC = the whole C matrix (conv: this image's [M][oH*oW] block): bytes = (M - 1) * ldc * 4 + N * 4"""
        c, e, st = self.c, self.p.emit, self.s_t
        C_ = self.ka0.sub(4, 2)
        e("s_lshl_b32", self.s_ldc4, self.s_ldc, 2)
        if c.conv:
            img = self.s_img if c.cpers else s(3)
            e("s_mul_hi_u32", st[2], img, self.s_scr[12])
            e("s_mul_i32", st[0], img, self.s_scr[12])
            e("s_add_u32", self.srdC[0], C_[0], st[0])
            e("s_addc_u32", self.srdC[1], C_[1], st[2])
            e("s_and_b32", self.srdC[1], self.srdC[1], 0xffff)
        else:
            e("s_mov_b32", self.srdC[0], C_[0])
            e("s_and_b32", self.srdC[1], C_[1], 0xffff)
        e("s_sub_u32", st[0], self.s_M, 1)
        e("s_mul_i32", st[0], st[0], self.s_ldc4)
        if c.conv:
            e("s_lshl_b32", st[2], self.s_N, 2)
        else:
            # C[i][j] at i * ldc + j * csC (MatrixView, gemm_utils.nim:36-60): bytes = (M - 1) * ldc * 4 + (N - 1) * csC * 4 + 4
            e("s_load_dword", self.s_csC4, s(0, 2), KA_EPI + 12)
            e("s_waitcnt", lgkmcnt=0)
            e("s_max_u32", self.s_csC4, self.s_csC4, 1)
            e("s_lshl_b32", self.s_csC4, self.s_csC4, 2)
            e("s_sub_u32", st[2], self.s_N, 1)
            e("s_mul_i32", st[2], st[2], self.s_csC4)
            e("s_add_u32", st[2], st[2], 4)
        e("s_add_u32", self.srdC[2], st[0], st[2])
        e("s_mov_b32", self.srdC[3], 0x00020000)
        e("s_mul_i32", self.s_ldc20, self.s_ldc4, 5)

    def issue_loads_all(self):
        if self.c.conv:      # (B's gathers first, like the loop body: the two queues must carry the same order)
            self.run_ops([o for grp in self.conv_load_ops() for o in grp])
        for pi in range(self.c.NPA):
            self.load_A_piece(pi)
        if self.c.conv:
            return
        for pj in range(self.c.NPB):
            self.load_B_piece(pj)

    # ------------------------------------------------------------------ implicit-GEMM convolution: the B operand
    # B "matrix" [K = Cin*kH*kW][N = oH*oW] of image b is never materialised (conv2d_im2col.nim:62-87 builds it explicitly):
    # element (k, pixel) = input[c][oh*sH + kh - pH][ow*sW + kw - pW], k = (c*kH + kh)*kW + kw.  One wave-instruction gathers one
    # output pixel per lane of ONE k: lane l owns pixels n0 + 2l and n0 + 2l + 1 (two dword loads per piece).  The k part of the
    # address -- (c*H*W + kh*W + kw) * 4 -- is wave-uniform and rides in the load's SGPR offset; the pixel part is a per-lane
    # constant (oh*sH*W + ow*sW) * 4 -- or, where the tap falls into the zero padding (or the pixel lies beyond the image, or k
    # beyond K), the offset 0x80000000 that the bounds check rejects, so the load returns 0 without touching memory.  Which of
    # the two depends on the tap (kh, kw), which is wave-uniform but changes every K-tile: the kH*kW (+1: "nothing") per-lane
    # offset vectors live in LDS and ds_read_addtid_b32 (address = M0 + 4*lane, no VGPR, no VALU op) fetches the right one -- any
    # VALU op in the loop costs ~11 cycles of matrix-pipe time (profiles/r03/asm_probe_v4_fillers.jsonl).  Kernel size, strides and
    # padding are RUN-TIME values (round 6): they only shape the table (built once per tile by a scalar loop over the taps, the
    # four waves taking every fourth tap) and the scalar tap arithmetic; the loop still has no vector instruction.  Wave w owns
    # k = 8w .. 8w+7 of every K-tile: pairs (k, k+2) per lane exactly like the GEMM's pair mode.
    CONV_DELTA = (0, 2, 1, 3, 4, 6, 5, 7)      # piece i = 2*pair + j gathers k = 8w + delta: pairs (0,2) (1,3) (4,6) (5,7)

    def conv_setup(self, again=False):
        """per tile: the geometry, the lanes' pixels, the tap table, srdB, the tap state; again (Cfg.cpers, the switch to the next unit
        inside the K loop): the per-lane LDS write addresses -- the same for every tile -- are left alone"""
        c, p, e, t, st, scr = self.c, self.p, self.p.emit, self.vt, self.s_t, self.s_scr
        B_ = self.ka0.sub(2, 2)
        sH, sW, soW, spH, spW, sCin, sNpix, smagic, sgeo, snt = (scr[i] for i in range(10))
        e("s_load_dwordx8", scr.sub(0, 8), s(0, 2), KA_CONV0)
        e("s_load_dwordx4", scr.sub(8, 4), s(0, 2), KA_CONV1)
        e("s_load_dwordx2", scr.sub(12, 2), s(0, 2), KA_CONV2)
        e("s_waitcnt", lgkmcnt=0)
        # geometry word: kH | kW << 8 | strideH << 16 | strideW << 24; taps word: kH * kW | ceil(1024 / kW) << 16
        e("s_bfe_u32", self.s_kW, sgeo, (8 << 16) | 8)
        e("s_and_b32", self.s_NT, snt, 0xffff)
        e("s_lshr_b32", self.s_M10, snt, 16)
        e("s_bfe_u32", scr[14], sgeo, (8 << 16) | 16)         # strideH
        e("s_lshr_b32", scr[15], sgeo, 24)                    # strideW
        sSH, sSW = scr[14], scr[15]
        lane, tab = t[0], t[1]
        pix = [t[2], t[3]]
        rowb, colb, base = [self.vT[0][0], self.vT[0][1]], [self.vT[0][2], self.vT[0][3]], [self.vT[0][4], self.vT[0][5]]
        e("v_and_b32", lane, 63, v(0))
        e("v_lshlrev_b32", tab, 2, lane)
        e("s_cmp_eq_u32", smagic, 0)
        e("s_cselect_b64", self.s_m, -1, 0)
        for ee in range(2):
            oh, ow = t[4], t[5]
            e("v_lshl_add_u32", pix[ee], lane, 1, self.s_n0)      # this lane's output pixel ee: n0 + 2 * lane + ee
            if ee:
                e("v_add_u32", pix[ee], 1, pix[ee])
            e("v_mul_hi_u32", oh, pix[ee], smagic)                # oh = pix / oW (the pair may straddle two output rows: odd widths)
            e("v_cndmask_b32", oh, oh, pix[ee], self.s_m)         # (oW == 1 travels as magic 0: oh = pix)
            e("v_mul_lo_u32", t[6], oh, soW)
            e("v_sub_u32", ow, pix[ee], t[6])                     # ow = pix % oW
            e("v_mul_lo_u32", t[6], oh, sSH)                      # oh * strideH
            e("v_mul_lo_u32", t[7], ow, sSW)                      # ow * strideW
            e("v_subrev_u32", rowb[ee], spH, t[6])                # input row of tap row 0 (as unsigned: negative = huge)
            e("v_subrev_u32", colb[ee], spW, t[7])
            e("v_mul_lo_u32", t[6], t[6], sW)
            e("v_add_u32", t[6], t[6], t[7])
            e("v_lshlrev_b32", base[ee], 2, t[6])                 # (oh*sH*W + ow*sW) * 4, relative to the window origin of pixel (0, 0)
            e("v_cmp_gt_u32", self.s_pok[ee], sNpix, pix[ee])     # pixels beyond the image (ragged last tile) gather nothing
        e("s_nop", 1)
        # the table: entry r = kh * kW + kw holds, per lane and pixel, the base offset where the tap reads inside the image, the
        # out-of-bounds offset elsewhere.  The same for the 4 waves (they own different k of the same 128 pixels): wave w writes the
        # entries w, w + 4, ...; wave 0 also the "nothing" entry (index taps) that channels beyond Cin select
        L_tap, L_tapd, L_non = p.label("tap"), p.label("tapsdone"), p.label("nonothing")
        sr, skh, skw, soff = st[0], st[2], st[3], st[4]
        e("s_mov_b32", sr, self.s_wave)
        p.place(L_tap)
        e("s_cmp_ge_u32", sr, self.s_NT)
        e("s_cbranch_scc1", L_tapd)
        e("s_mul_i32", skh, sr, self.s_M10)
        e("s_lshr_b32", skh, skh, 10)                             # kh = r / kW
        e("s_mul_i32", skw, skh, self.s_kW)
        e("s_sub_u32", skw, sr, skw)                              # kw = r % kW
        e("s_lshl_b32", soff, sr, 8)
        e("v_add_u32", t[6], soff, tab)
        for ee in range(2):
            e("v_add_u32", t[4], skh, rowb[ee])
            e("v_cmp_gt_u32", VCC, sH, t[4])                      # input row oh*sH + kh - pH exists
            e("v_add_u32", t[5], skw, colb[ee])
            e("v_cmp_gt_u32", self.s_m, sW, t[5])                 # input column exists
            e("s_nop", 1)
            e("s_and_b64", self.s_m, self.s_m, VCC)
            e("s_and_b64", self.s_m, self.s_m, self.s_pok[ee])
            e("s_nop", 0)
            e("v_cndmask_b32", t[7], self.v_oob, base[ee], self.s_m)
            e("ds_write_b32", t[6], t[7], offset=ee * c.TAB_E1)
        e("s_add_u32", sr, sr, 4)
        e("s_branch", L_tap)
        p.place(L_tapd)
        e("s_cmp_lg_u32", self.s_wave, 0)
        e("s_cbranch_scc1", L_non)
        e("s_lshl_b32", soff, self.s_NT, 8)
        e("v_add_u32", t[6], soff, tab)
        e("ds_write_b32", t[6], self.v_oob)
        e("ds_write_b32", t[6], self.v_oob, offset=c.TAB_E1)
        p.place(L_non)
        e("s_waitcnt", lgkmcnt=0)
        e("s_barrier")
        # descriptor: base = B + b * bsB - (pH*W + pW) * 4 (the window origin of output pixel (0, 0), kernel tap (0, 0));
        # every address a valid lane forms lies inside the image -- the bounds field only has to reject v_oob
        img = self.s_img if c.cpers else s(3)
        e("s_mul_hi_u32", st[2], img, scr[10])
        e("s_mul_i32", st[0], img, scr[10])
        e("s_add_u32", st[0], B_[0], st[0])
        e("s_addc_u32", st[2], B_[1], st[2])
        e("s_mul_i32", st[3], spH, sW)
        e("s_add_u32", st[3], st[3], spW)
        e("s_lshl_b32", st[3], st[3], 2)
        e("s_sub_u32", self.srdB[0], st[0], st[3])
        e("s_subb_u32", self.srdB[1], st[2], 0)
        e("s_and_b32", self.srdB[1], self.srdB[1], 0xffff)
        e("s_mov_b32", self.srdB[2], 0x7fffffff)
        e("s_lshl_b32", self.s_W4, sW, 2)
        e("s_mul_i32", self.s_HW4, self.s_W4, sH)
        e("s_mov_b32", self.s_Cin, sCin)
        # LDS write addresses of pair gi, pixel e: x = 2*lane + e, k = 8w + (0, 1, 4, 5)[gi]: L = 2w + (gi & 1), word = (0,0,2,2)[gi]
        e("s_mov_b32", st[0], c.LDS0 + c.ROWP * c.BM)
        for gi in range(0 if again else 4):
            e("s_lshl_b32", st[3], self.s_wave, 1)
            e("s_add_u32", st[3], st[3], gi & 1)
            for ee in range(2):
                xx, rr, ss = t[4], t[5], t[6]
                e("v_lshl_add_u32", xx, lane, 1, ee)
                self.kq_row(rr, xx, t[7])
                self.kq_swz(ss, xx, t[7])
                e("v_xor_b32", ss, st[3], ss)
                e("v_mul_u32_u24", rr, c.ROWP, rr)
                e("v_lshl_add_u32", rr, ss, 4, rr)
                e("v_add_u32", rr, 4 * (0, 0, 2, 2)[gi], rr)
                e("v_add_u32", self.lds_at(self.WB[gi][ee][2])[0], st[0], rr)
                if c.il:
                    continue
                e("v_add_u32", self.WB[gi][ee][0], c.STAGE, self.WB[gi][ee][2])
                e("v_add_u32", self.WB[gi][ee][1], 2 * c.STAGE, self.WB[gi][ee][2])
        # running k of this wave: 8w, + BK per K-tile
        e("s_lshl_b32", self.s_k0, self.s_wave, 3)

    def conv_load_ops(self):
        """per piece: [scalar tap arithmetic] [SGPR offset, table entry -> M0, the two offset vectors from LDS]; then, for every
        piece, [the two gathers]; then the state moves on by BK.  Returns a list of op groups (one per MFMA gap).
        k -> channel c = k / taps (magic multiply: KA_TAB carries floor(2^32 / taps) + 1, 0 for one tap), r = k % taps,
        kh = r / kW = (r * ceil(1024 / kW)) >> 10 (exact for r < 49), kw = r % kW."""
        c, st = self.c, self.s_t
        mgNT = self.ka0[6]
        groups, loads = [], []
        for i in range(8):
            d = self.CONV_DELTA[i]
            g1 = [("ins", "s_add_u32", (st[0], self.s_k0, d), {}),               # k
                  ("ins", "s_mul_hi_u32", (st[1], st[0], mgNT), {}),
                  ("ins", "s_cmp_eq_u32", (mgNT, 0), {}),
                  ("ins", "s_cselect_b32", (st[1], st[0], st[1]), {}),           # c = k / taps
                  ("ins", "s_mul_i32", (st[2], st[1], self.s_NT), {}),
                  ("ins", "s_sub_u32", (st[0], st[0], st[2]), {}),               # r = k % taps
                  ("ins", "s_mul_i32", (st[2], st[0], self.s_M10), {}),
                  ("ins", "s_lshr_b32", (st[2], st[2], 10), {}),                 # kh = r / kW
                  ("ins", "s_mul_i32", (st[3], st[2], self.s_kW), {}),
                  ("ins", "s_sub_u32", (st[5], st[0], st[3]), {})]               # kw = r % kW
            g2 = [("ins", "s_mul_i32", (st[3], st[1], self.s_HW4), {}),
                  ("ins", "s_mul_i32", (st[4], st[2], self.s_W4), {}),
                  ("ins", "s_add_u32", (st[3], st[3], st[4]), {}),
                  ("ins", "s_lshl_b32", (st[4], st[5], 2), {}),
                  ("ins", "s_add_u32", (self.s_koff[i], st[3], st[4]), {}),  # (c*H*W + kh*W + kw) * 4
                  ("ins", "s_cmp_lt_u32", (st[1], self.s_Cin), {}),
                  ("ins", "s_cselect_b32", (st[0], st[0], self.s_NT), {}),   # channels beyond Cin (k >= K): the "nothing" entry
                  ("ins", "s_lshl_b32", (M0, st[0], 8), {}),
                  ("ins", "s_nop", (0,), {}),                                # (S_MOV to M0 -> LDS add-TID instruction: 1 wait state)
                  ("ldsr", "ds_read_addtid_b32", (self.vB0[i],), {}, ("T", i)),
                  ("ldsr", "ds_read_addtid_b32", (self.vB1[i],), {"offset": c.TAB_E1}, ("T", i))]
            groups += [g1, g2]
            loads.append([("lgwait", {("T", i)}), ("loadBc", i, 0), ("loadBc", i, 1)])
        groups += loads
        groups.append([("ins", "s_add_u32", (self.s_k0, self.s_k0, c.BK), {})])
        return groups

    def conv_store_ops(self, k):
        """tile data in the B staging registers -> LDS stage index k; returns op groups"""
        out = []
        for gi in range(4):
            P, Q = self.stB[2 * gi], self.stB[2 * gi + 1]
            for ee in range(2):
                g = [("vmwait", ("B", 2 * gi + 1, 1))] if ee == 0 else []
                g.append(self.w2(self.WB[gi][ee][k], P[ee], Q[ee]))
                out.append(g)
        return out

    def load_A_piece(self, pi):
        if "loads" in self.c.ablate:
            return
        self.p.emit("buffer_load_dwordx4", self.stA[pi], self.vVA[pi], self.srdA, 0, offen=True)
        self.vm_issue(("A", pi))

    def load_B_piece(self, pj):
        if "loads" in self.c.ablate:
            return
        self.p.emit("buffer_load_dwordx4", self.stB[pj], self.vVB[pj], self.srdB, 0, offen=True)
        self.vm_issue(("B", pj))

    def advance_srds(self, which=None):
        """descriptors move one K-tile along k: base += step, bytes = max(bytes - step, 0)"""
        e = self.p.emit
        ops = []
        for srd, step in ((self.srdA, self.c.BK * 4),) + (() if self.c.conv else ((self.srdB, self.s_bstep),)):
            ops += [("s_add_u32", srd[0], srd[0], step), ("s_addc_u32", srd[1], srd[1], 0),
                    ("s_sub_u32", srd[2], srd[2], step), ("s_cselect_b32", srd[2], 0, srd[2])]
        if which is None:
            for o in ops:
                e(*o)
        return ops

    def pre_op(self, regs, mask):
        """fused prologue: the staged elements become (mask ? max(x, 0) : x) before they are stored -- all of a piece's VALU work in one
        gap (the first VALU instruction of a gap costs ~11 cycles of matrix-pipe time, every further one ~4)"""
        if not self.c.pre:
            return []
        tmp = [self.vt[4 + j] for j in range(4)]       # (free in the K loop: the fold uses vt[0..3] and vT)

        def emit():
            e = self.p.emit
            for j, r in enumerate(regs):
                e("v_max_f32", tmp[j % 4], 0, r)
            for j, r in enumerate(regs):
                e("v_cndmask_b32", r, r, tmp[j % 4], mask)
        assert len(regs) <= 4
        return [("call", emit)]

    def w2(self, entry, d0, d1):
        """two adjacent words of one stage's panel image from two registers (entry = that stage's slot of a write-address triple)"""
        reg, off = self.lds_at(entry)
        return ("ldsw", "ds_write2_b32", (reg, d0, d1), {"offset0": off // 4, "offset1": off // 4 + 1})

    def store_A_piece(self, pi, ops=None, k=0):
        """piece = 4 consecutive k (e0 e1 e2 e3) of one row: (e0, e2) -> chunk of MFMA half 0, (e1, e3) -> half 1"""
        out = []
        r = self.stA[pi]
        out.append(("vmwait", ("A", pi)))
        out += self.pre_op([r[j] for j in range(4)], self.s_preA)
        # ds_write2_b32 takes its two dwords from two independent registers: no repacking VALU op (a v_swap on freshly
        # loaded registers cost 16 cycles of matrix-pipe time per piece, profiles/r03/asm_probe_v7.jsonl); the price is one
        # address register per (piece, half, stage)
        out.append(self.w2(self.WA[0][pi][k], r[0], r[2]))
        out.append(self.w2(self.WA[1][pi][k], r[1], r[3]))
        if ops is None:
            self.run_ops(out)
        return out

    def store_B_kpiece(self, pj, ops=None, k=0):
        """B passed transposed (k-contiguous): the same two ds_write2_b32 per piece as A"""
        out = []
        r = self.stB[pj]
        out.append(("vmwait", ("B", pj)))
        out += self.pre_op([r[j] for j in range(4)], self.s_preB)
        out.append(self.w2(self.WB[0][pj][k], r[0], r[2]))
        out.append(self.w2(self.WB[1][pj][k], r[1], r[3]))
        if ops is None:
            self.run_ops(out)
        return out

    def apply_tail_mask(self):
        """the loads issued from here on fetch the last K-tile: pieces beyond K read as 0"""
        e = self.p.emit
        regs = list(self.vVA) + (list(self.vVB) if (self.c.b_kcontig and not self.c.conv) else [])
        for r in regs:
            e("v_cndmask_b32", r, self.v_oob, r, self.s_tm)

    def mask_last_pieces_if(self, sreg, value):
        """K % 4 != 0 and `sreg` == value: the staging registers hold the last K-tile (requested one step earlier): wait for it and
        zero the elements beyond K.  ~NPA * 4 v_cndmask once per workgroup; the counted waits that follow are then satisfied early."""
        c, e = self.c, self.p.emit
        if c.conv:
            return
        skip = self.p.label("nok4")
        e("s_and_b32", self.s_t[0], self.s_Keff, 3)
        e("s_cmp_eq_u32", self.s_t[0], 0)
        e("s_cbranch_scc1", skip)
        e("s_cmp_lg_u32", sreg, value)
        e("s_cbranch_scc1", skip)
        e("s_waitcnt", vmcnt=0)
        for r in list(self.stA) + (list(self.stB) if c.b_kcontig else []):
            for j in range(4):
                if r.kind == "a":      # (runv: the staging registers are AGPRs -- through a temporary; once per run, K % 4 != 0 only)
                    tmp = self.vt[4 + j]
                    e("v_accvgpr_read_b32", tmp, r[j])
                    e("v_cndmask_b32", tmp, 0, tmp, self.s_em[j])
                    e("v_accvgpr_write_b32", r[j], tmp)
                    continue
                e("v_cndmask_b32", r[j], 0, r[j], self.s_em[j])
        self.p.place(skip)

    def tail_mask_if(self, sreg, value):
        """apply_tail_mask when K has a tail and `sreg` == value (uniform branch around ~12 VALU instructions, once per workgroup)"""
        e = self.p.emit
        skip = self.p.label("notail")
        e("s_cmp_eq_u32", self.s_ktail, 0)
        e("s_cbranch_scc1", skip)
        e("s_cmp_lg_u32", sreg, value)
        e("s_cbranch_scc1", skip)
        self.apply_tail_mask()
        self.p.place(skip)

    def store_B_pair(self, gi, ops=None, k=0):
        """pieces P (row k) and Q (row k + 2) of one x quad: element e of both -> adjacent words of row x = 4xq + e"""
        out = []
        P, Q = self.stB[2 * gi], self.stB[2 * gi + 1]
        out.append(("vmwait", ("B", 2 * gi + 1)))
        out += self.pre_op([P[j] for j in range(4)], self.s_preB)
        out += self.pre_op([Q[j] for j in range(4)], self.s_preB)
        if self.c.b_store == "write2":
            for ee in range(4):
                out.append(self.w2(self.WB[gi][ee][k], P[ee], Q[ee]))
        else:
            # (P0 P1 P2 P3 Q0 Q1 Q2 Q3) -> (P0 Q0 P1 Q1 P2 Q2 P3 Q3): two 3-cycles = 4 swaps; P, Q are adjacent registers
            assert Q.idx == P.idx + 4
            r = [v(P.idx + k) for k in range(8)]
            for x, y in ((1, 4), (2, 4), (3, 5), (6, 5)):
                out.append(("ins", "v_swap_b32", (r[x], r[y]), {}))
            for ee in range(4):
                out.append(("ldsw", "ds_write_b64", (self.WB[gi][ee][k], v(P.idx + 2 * ee, 2)), {}))
        if ops is None:
            self.run_ops(out)
        return out

    def run_ops(self, ops):
        for o in ops:
            self.run_op(o)

    def run_op(self, o):
        kind = o[0]
        ab = self.c.ablate
        if (kind == "ldsw" and "stores" in ab) or (kind == "ldsr" and "reads" in ab) or (kind in ("loadA", "loadB") and "loads" in ab):
            return
        if kind == "ins" and o[1] == "v_swap_b32" and ("stores" in ab or "swaps" in ab):
            return
        if kind == "ldsw" and ((o[1] == "ds_write_b64" and "awrites" in ab) or (o[1] == "ds_write2_b32" and "bwrites" in ab)):
            return
        if kind == "barrier" and "barrier" in ab:
            self.lg_wait(None)
            return
        if kind == "vmwait":
            self.vm_wait(o[1])
        elif kind == "lgwait":
            self.lg_wait(o[1])
        elif kind == "ins":
            self.p.emit(o[1], *o[2], **o[3])
        elif kind == "ldsw":
            self.p.emit(o[1], *o[2], **o[3])
            self.lg_issue(("W",))
        elif kind == "ldsr":
            self.p.emit(o[1], *o[2], **o[3])
            self.lg_issue(o[4])
        elif kind == "loadA":
            self.load_A_piece(o[1])
        elif kind == "loadB":
            self.load_B_piece(o[1])
        elif kind == "loadBc":
            if "loads" not in self.c.ablate:
                self.p.emit("buffer_load_dword", self.stB[o[1]][o[2]], (self.vB0, self.vB1)[o[2]][o[1]], self.srdB, self.s_koff[o[1]], offen=True)
                self.vm_issue(("B", o[1], o[2]))
        elif kind == "barrier":
            self.lg_wait(None)
            self.p.emit("s_barrier")
        elif kind == "call":
            o[1]()
        else:
            raise ValueError(kind)

    def read_group_ops(self, g, stage, slot):
        """fragment reads of group g (4 k-steps) into register slot `slot`; stage 0: the tile being multiplied, 1: the next tile"""
        c = self.c
        ops = []
        blk = c.MB * c.ROWP
        ra, oa = self.lds_at(self.RA[g][stage])
        rb, ob = self.lds_at(self.RB[g][stage])
        for i in range(c.TM):
            ops.append(("ldsr", "ds_read_b128", (self.fa[slot][i], ra), {"offset": i * blk + oa}, ("R", slot)))
        for n in range(c.TN):
            ops.append(("ldsr", "ds_read_b128", (self.fb[slot][n], rb), {"offset": n * blk + ob}, ("R", slot)))
        return ops

    def read_group(self, g, stage, slot):
        self.run_ops(self.read_group_ops(g, stage, slot))

    def staging_ops(self, wr_k):
        """LDS stores of tile t+1 (from registers), each followed by the HBM load of tile t+2 into the drained registers"""
        c = self.c
        stg = []
        for pi in range(c.NPA):
            stg += self.store_A_piece(pi, ops=[], k=wr_k)
            stg.append(("loadA", pi))
        if c.conv:
            pass
        elif c.b_kcontig:
            for pj in range(c.NPB):
                stg += self.store_B_kpiece(pj, ops=[], k=wr_k)
                stg.append(("loadB", pj))
        else:
            for gi in range(c.NPB // 2):
                stg += self.store_B_pair(gi, ops=[], k=wr_k)
                stg.append(("loadB", 2 * gi))
                stg.append(("loadB", 2 * gi + 1))
        return stg

    # ------------------------------------------------------------------ the matrix instruction and the slice fold
    def emit_mfma(self, b, slot, i, n, u, srcc):
        self.p.emit("v_mfma_f32_32x32x2_f32", self.acc[b], self.fa[slot][i][u], self.fb[slot][n][u], srcc)

    def fold_before(self, b):
        # Laser's pc loop (gemm.nim:150-158): the finished slice sum leaves the accumulator (16 reads in the shadow of
        # the previous MFMA), this MFMA starts the next chain from a literal +0, and run += slice follows it -- all
        # 64 VALU operations of a block in ONE gap: beside the f32 MFMA stream the first VALU instruction of a gap
        # costs ~11 cycles of matrix-pipe time and every further one ~4 (profiles/r03/asm_probe_v4_fillers.jsonl), so
        # VALU work is batched, never spread.  (Moving the sums AGPR -> LDS -> VGPR instead, which needs no VALU
        # moves, measured slower: profiles/r03/asm_probe_v8.jsonl "exact_lds".)
        T = self.vT[0]
        if "foldreads" in self.c.ablate:      # (pricing experiments: results are wrong)
            return
        for r in range(16):
            self.p.emit("v_accvgpr_read_b32", T[r], self.acc[b][r])

    def fold_after(self, b):
        p, e, T = self.p, self.p.emit, self.vT[0]
        # run += alpha * slice, unfused (gemm_ukernel_generic.nim:68-76); alpha == 1 (every reference caller): 1*x is x;
        # the multiplies sit out of line (after s_endpgm) so that the usual case is a branch NOT taken
        lmul, lback = p.label("amul"), p.label("aback")
        e("s_cmp_lg_u32", self.s_alpha, 0x3f800000)
        e("s_cbranch_scc1", lmul)
        p.place(lback)
        self.outlined.append((lmul, [("v_mul_f32", T[r], self.s_alpha, T[r]) for r in range(16)], lback))
        if c_runv(self):
            # the running sum is in arch VGPRs: two adds per instruction, nothing to move (8 VALU operations where the AGPR plan has 48)
            if "foldadds" in self.c.ablate:
                return
            if "foldscalar" in self.c.ablate:     # (pricing: 16 single adds instead of 8 packed ones -- same result)
                for r in range(16):
                    e("v_add_f32", self.run[b][r], self.run[b][r], T[r])
                return
            for j in range(8):
                e("v_pk_add_f32", self.run[b].sub(2 * j, 2), self.run[b].sub(2 * j, 2), T.sub(2 * j, 2))
            return
        for r in range(16):
            tt = self.vt[r % 4]
            e("v_accvgpr_read_b32", tt, self.run[b][r])
            e("v_add_f32", tt, tt, T[r])
            e("v_accvgpr_write_b32", self.run[b][r], tt)

    def trans_after(self, b):
        """transition body (Cfg.pipe), block b: vT holds alpha-less slice sum of the tile being FINISHED (fold_before), the MFMA in front of
        this gap has restarted the chain for the new tile.  Laser-order: C = run + alpha * slice -- the last fold and the epilogue's
        sum are the same operation (gemm.nim:150-158, gemm_ukernel_generic.nim:68-76) -- stored straight from the temporaries, and
        the running sum is zeroed for the new tile (beta == 0 in pipelined launches).  One chain: C = alpha * sum.  The rows of a
        block advance through the store's SCALAR offset, which gfx950 includes in the range check of a raw buffer (probe:
        scripts/probes/buffer_soffset.hip -- voffset + soffset + 4 <= num_records): rows beyond M fall off the end of srdCd, columns
        beyond N carry an out-of-range vC -- dropped exactly as in the epilogue.  No vector address arithmetic."""
        c, p, e, T, st = self.c, self.p, self.p.emit, self.vT[0], self.s_t
        i, n = b // c.TN, b % c.TN
        lmul, lback = p.label("tmul"), p.label("tback")
        e("s_cmp_lg_u32", self.s_alpha, 0x3f800000)
        e("s_cbranch_scc1", lmul)
        p.place(lback)
        self.outlined.append((lmul, [("v_mul_f32", T[r], self.s_alpha, T[r]) for r in range(16)], lback))
        soff = st[0]
        if i:
            e("s_mul_i32", soff, self.s_ldc4, 32 * i)
        else:
            e("s_mov_b32", soff, 0)
        blk4 = v(self.vt[0].idx, 4)      # vt[0..3]: two register pairs
        for r in range(16):
            rr = r & 3
            if c.runv:
                # run in arch VGPRs: C = run + alpha * slice two elements at a time into a temporary pair, the pair of the running sum zeroed
                pair = blk4.sub(2 * ((r >> 1) & 1), 2)
                if r % 2 == 0:
                    e("v_pk_add_f32", pair, self.run[b].sub(r, 2), T.sub(r, 2))
                    e("v_mov_b64", self.run[b].sub(r, 2), 0)
                tt = pair[r & 1]
            elif c.exact:
                tt = self.vt[r % 4]
                e("v_accvgpr_read_b32", tt, self.run[b][r])
                e("v_add_f32", tt, tt, T[r])
                e("v_accvgpr_write_b32", self.run[b][r], 0)
            else:
                tt = T[r]
            if "cstores" not in c.ablate:
                e("buffer_store_dword", tt, self.vC[n], self.srdCd, soff, offen=True)
                self.vm_issue(("S", b, r))
            if r < 15:
                e("s_add_u32", soff, soff, self.s_ldc4 if rr < 3 else self.s_ldc20)

    def pipe_c_addr(self):
        """(vC, srdCd) for the deferred stores of the tile (m0, n0): srdCd starts at this wave's first row, vC[n] = the lane's
        offset from there ((4 * hi) rows down, block column n; out of bounds beyond N)"""
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        lane, lo, hi = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, v(0))
        e("v_and_b32", lo, 31, lane)
        e("v_lshrrev_b32", hi, 5, lane)
        e("s_add_u32", st[0], self.s_m0, self.s_wm0)
        e("s_mul_hi_u32", st[2], st[0], self.s_ldc4)
        e("s_mul_i32", st[3], st[0], self.s_ldc4)
        e("s_add_u32", self.srdCd[0], self.srdC[0], st[3])
        e("s_addc_u32", self.srdCd[1], self.srdC[1], st[2])
        e("s_and_b32", self.srdCd[1], self.srdCd[1], 0xffff)
        e("s_mov_b32", self.srdCd[3], 0x00020000)
        e("s_sub_u32", self.srdCd[2], self.srdC[2], st[3])              # what is left of C from there (0: the wave's rows lie beyond M)
        e("s_cselect_b32", self.srdCd[2], 0, self.srdCd[2])
        e("s_cmp_lg_u32", st[2], 0)
        e("s_cselect_b32", self.srdCd[2], 0, self.srdCd[2])
        e("v_lshlrev_b32", t[3], 2, hi)
        e("v_mul_lo_u32", t[3], t[3], self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], lo)
        if self.s_csC4 is None:           # (convolution kernels: C's columns are adjacent)
            e("v_lshl_add_u32", t[3], t[4], 2, t[3])
            e("s_mov_b32", st[5], 128)
        else:
            e("v_mul_lo_u32", t[6], t[4], self.s_csC4)
            e("v_add_u32", t[3], t[3], t[6])
            e("s_lshl_b32", st[5], self.s_csC4, 5)
        for n in range(c.TN):
            e("v_add_u32", t[5], 32 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            if n == 0:
                e("v_mov_b32", t[6], t[3])
            else:
                e("v_add_u32", t[6], st[5], t[6])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[6], VCC)

    def next_unit(self, L_none):
        """Cfg.cpers: the workgroup's next unit -> s_img, s_m0, s_n0 (L_none when there is none).  KA_SCHED2 carries, for these kernels,
        +4 the tiles of one image, +8 their magic number, +12 the stride (= workgroups), +16 the units of the launch (images x tiles).
        Clobbers s_scr[0..7] (dead outside a tile body's load section) and s_t[0..5]."""
        c, e, st, sc = self.c, self.p.emit, self.s_t, self.s_sc
        e("s_load_dwordx8", sc, s(0, 2), KA_SCHED2)
        e("s_waitcnt", lgkmcnt=0)
        e("s_cmp_ge_u32", self.s_tcur, sc[4])
        e("s_cbranch_scc1", L_none)
        self.udiv(self.s_img, self.s_tcur, sc[2])
        e("s_mul_i32", st[4], self.s_img, sc[1])
        e("s_sub_u32", st[4], self.s_tcur, st[4])                     # the tile inside the image
        e("s_add_u32", self.s_tcur, self.s_tcur, sc[3])
        e("s_load_dwordx8", sc, s(0, 2), KA_SCHED)
        e("s_waitcnt", lgkmcnt=0)
        self.tile_coords(st[4], sc, st[0], st[1], (st[2], st[3], st[5]))
        e("s_mul_i32", self.s_m0, st[0], c.BM)
        e("s_mul_i32", self.s_n0, st[1], c.BN)

    def pipe_switch_conv(self, stubs, rets):
        """pipe_switch of the convolution kernels (Cfg.cpers).  Every wave that gets here has passed the barrier of the tile body it
        comes from, and a wave only reaches that barrier after its last look at the tap table (the table reads of a body precede its
        gathers, the gathers the barrier): the table of the tile being finished can be overwritten at once; conv_setup's own barrier
        stands between the new table's writes and its first reads.  Scalar tap state, srdB and srdC move to the next unit; the C
        addresses of the tile being finished were put aside first."""
        c, p = self.c, self.p
        e, st = p.emit, self.s_t
        L_sw, L_out = p.label("switch"), p.label("swout")
        for site, lab in stubs.items():
            p.place(lab)
            e("s_or_b32", self.s_pipe, self.s_pipe, site << 4)
            e("s_branch", L_sw)
        p.place(L_sw)
        e("s_bitcmp1_b32", self.s_pipe, 0)
        e("s_cbranch_scc0", L_out)
        e("s_load_dword", st[0], s(0, 2), KA_SCHED2 + 16)
        e("s_waitcnt", lgkmcnt=0)
        e("s_cmp_lt_u32", self.s_tcur, st[0])
        e("s_cbranch_scc0", L_out)
        self.pipe_c_addr()
        self.next_unit(L_out)
        self.ab_descriptors(again=True)
        self.c_descriptor()
        e("s_or_b32", self.s_pipe, self.s_pipe, 2)
        p.place(L_out)
        e("s_lshr_b32", st[0], self.s_pipe, 4)
        e("s_and_b32", self.s_pipe, self.s_pipe, 15)
        sites = sorted(rets)
        for site in sites[:-1]:
            e("s_cmp_eq_u32", st[0], site)
            e("s_cbranch_scc1", rets[site])
        e("s_branch", rets[sites[-1]])

    def pipe_switch(self, stubs, rets):
        if self.c.cpers:
            return self.pipe_switch_conv(stubs, rets)
        """out of line, reached from the tail of the tile body after which TWO K-tiles of the tile are left (every load of the tile has
        been requested): if the next run of this workgroup is another whole tile of a launch that may pipeline, the C addresses of the
        tile being finished are put aside, the scheduler moves on and srdA / srdB are pointed at the next tile -- the two tile bodies
        that follow fetch ITS first two K-tiles where they would have fetched zeros past K.  stubs[site] set the return site."""
        c, p = self.c, self.p
        e, st, sc = p.emit, self.s_t, self.s_sc
        L_sw, L_out = p.label("switch"), p.label("swout")
        for site, lab in stubs.items():
            p.place(lab)
            e("s_or_b32", self.s_pipe, self.s_pipe, site << 4)
            e("s_branch", L_sw)
        p.place(L_sw)
        e("s_bitcmp1_b32", self.s_pipe, 0)
        e("s_cbranch_scc0", L_out)
        e("s_cmp_eq_u32", self.s_phase, 1)                  # (the head slices of phase 0 run with s_phase == 1 too: END_SEND tells)
        e("s_cbranch_scc0", L_out)
        e("s_cmp_eq_u32", self.s_end, END_EPI)
        e("s_cbranch_scc0", L_out)
        e("s_cmp_lt_u32", self.s_tcur, self.s_t1)
        e("s_cbranch_scc0", L_out)
        self.pipe_c_addr()
        e("s_load_dwordx8", sc, s(0, 2), KA_SCHED2)
        e("s_waitcnt", lgkmcnt=0)
        self.tile_step(st[1])
        e("s_mov_b32", self.s_tile, self.s_tcur)
        e("s_add_u32", self.s_tcur, self.s_tcur, st[1])
        e("s_load_dwordx8", sc, s(0, 2), KA_SCHED)
        e("s_waitcnt", lgkmcnt=0)
        e("s_add_u32", st[4], self.s_tile, self.ka0[6])           # (the launch's first tile: prologue)
        self.tile_coords(st[4], sc, st[0], st[1], (st[2], st[3], st[5]))
        e("s_mul_i32", self.s_m0, st[0], c.BM)
        e("s_mul_i32", self.s_n0, st[1], c.BN)
        self.ab_descriptors()
        e("s_or_b32", self.s_pipe, self.s_pipe, 2)
        p.place(L_out)
        e("s_lshr_b32", st[0], self.s_pipe, 4)
        e("s_and_b32", self.s_pipe, self.s_pipe, 15)
        sites = sorted(rets)
        for site in sites[:-1]:
            e("s_cmp_eq_u32", st[0], site)
            e("s_cbranch_scc1", rets[site])
        e("s_branch", rets[sites[-1]])

    # ------------------------------------------------------------------ one K-tile
    def tile_body(self, fold, stage=None, regset=0, trans=False):
        """one K-tile.  fold: the first tile of a kc slice (the slice fold rides in its first gaps).  trans (Cfg.pipe): the first K-tile
        of a tile whose predecessor in this workgroup has not been stored yet: a fold tile whose fold finishes the OLD tile -- every
        block's C values leave for memory from the gap behind the MFMA that restarts its chain (trans_after)."""
        c, p = self.c, self.p
        e = p.emit
        assert fold or not trans
        self.stA, self.stB = self.st_sets[regset]      # the set this body drains into LDS and refills
        gaps = {m: [] for m in range(-1, c.NMF)}   # gap m: ops issued right after MFMA m (gap -1: before MFMA 0)

        def put(m, op):
            gaps[min(max(m, -1), c.NMF - 1)].append(op)

        # (a fold tile carries the slice fold in its first gaps: its staging starts later, so its barrier sits late)
        bar = max(c.bar_gap, c.NMF - c.GM - 1) if fold else c.bar_gap
        if fold and bar - 1 - (c.NB + 1 + c.TM + c.TN + 2) < 2 * c.NPA + 3 * c.NPB:   # small tiles: as late as the prefetch allows
            bar = c.NMF - (c.TM + c.TN) * c.r_step - 3
        first_free = 0
        if fold:
            first_free = c.NB + 1           # gaps 0..NB-1 carry the slice fold
            if bar - 1 - (first_free + c.TM + c.TN + 2) < 2:      # very few gaps per tile: the barrier as late as the prefetch allows
                bar = max(bar, c.NMF - 3)
        # fragment reads: group g+1 (or group 0 of the next tile) during group g
        for g in range(c.NG):
            nxt_tile = (g + 1 == c.NG)
            which = (stage + 1) % 3 if nxt_tile else stage
            ops = self.read_group_ops(0 if nxt_tile else g + 1, which, (g + 1) & 1)
            base = g * c.GM + 1
            if nxt_tile:
                base = max(base, bar + 1)
            if fold and g == 0:
                base = first_free
            for k, op in enumerate(ops):
                at = base + k * c.r_step
                if not nxt_tile:
                    at = min(at, max(base, (g + 1) * c.GM - 2))      # (few gaps per group: several reads share a gap)
                assert nxt_tile or at < (g + 1) * c.GM - 1, "fragment reads must be issued before the group's wait"
                put(at, op)
            # the group's fragments must have landed before its first MFMA (the next tile's first group: top of the body)
            if not nxt_tile:
                put((g + 1) * c.GM - 1, ("lgwait", {("R", (g + 1) & 1)}))
        # staging: LDS stores of tile t+1 (from registers), then the HBM loads of tile t+2 into the drained registers
        # register of a triple: [0] / [1] = this tile's / the next tile's stage when the triples rotate; with one body per
        # stage, reads of stage k use index k and writes index (k + 2) % 3 (the triples were initialised for stage 0)
        wr_k = ((stage + 1) % 3 + 2) % 3
        stg = self.staging_ops(wr_k)
        # waits ride with the op that follows them
        units = []
        for op in stg:
            if units and (units[-1][-1][0] in ("vmwait",) or (op[0] == "call" and units[-1][-1][0] == "call")):
                units[-1].append(op)        # (a wait, and a fused prologue's VALU batch, ride with the op that follows / precedes them)
            else:
                units.append([op])
        if c.conv:
            # B first (its gathers were requested a tile ago and are needed soonest): stores, then the scalar tap state and
            # the offset vectors of tile t+2, its 16 gathers, then A's pieces; one unit per gap
            cu = self.conv_store_ops(wr_k)
            a_units = []
            for pi in range(c.NPA):
                a = self.store_A_piece(pi, ops=[], k=wr_k)
                a_units += [a[:2], [a[2]], [("loadA", pi)]]
            units = cu + self.conv_load_ops() + a_units
        w0 = max(c.w_start, first_free + (c.TM + c.TN + 2 if fold else 0))
        if fold and bar - 1 - w0 < 2:         # very few gaps (16 MFMAs per tile): staging shares the gaps of the fragment reads
            w0 = first_free
        span = bar - 1 - w0
        step = c.w_step or min(max(1.0, span / max(1, len(units))), span / max(1, len(units) - 1) if len(units) > span else 1e9)
        for k, u in enumerate(units):
            m = int(w0 + k * step)
            assert m < bar, "staging does not fit before the barrier"
            for op in u:
                put(m, op)
        put(bar, ("barrier",))
        # scalar bookkeeping after the last load of the tile
        tail = bar + 2
        for k, o in enumerate(self.advance_srds(which="ops")):
            put(tail + k, ("ins", o[0], o[1:], {}))

        # MFMA order: k-step-major -- NB independent accumulators between two uses of one
        order = [(g, u, b) for g in range(c.NG) for u in range(c.KREAD) for b in range(c.NB)]

        if c.filler:
            nvm = 0
            for m_ in range(0, c.NMF, c.filler_every):
                f = c.filler
                if f in ("ds_write_b64", "ds_write_b128", "ds_write_b32"):
                    n_ = {"ds_write_b32": 1, "ds_write_b64": 2, "ds_write_b128": 4}[f]
                    put(m_, ("ins", f, (self.vFaddr, self.vF.sub(0, n_) if n_ > 1 else self.vF[0]), {}))
                elif f == "ds_write2_b32":
                    put(m_, ("ins", f, (self.vFaddr, self.vF[0], self.vF[2]), {"offset0": 0, "offset1": 1}))
                elif f == "ds_read_b128":
                    put(m_, ("ins", f, (self.vF, self.vFaddr), {}))
                elif f == "buffer_load_dwordx4":
                    put(m_, ("ins", f, (self.vF, self.vFoff, self.srdA, 0), {"offen": True}))
                    nvm += 1
                    if nvm % 16 == 0:
                        put(m_, ("ins", "s_waitcnt", (), {"vmcnt": 8}))
                elif f in ("v_swap_b32",):
                    put(m_, ("ins", f, (self.vF[0], self.vF[1]), {}))
                elif f == "v_swap_distinct":
                    put(m_, ("ins", "v_swap_b32", (self.vT[0][m_ % 16], self.vT[0][(m_ + 5) % 16]), {}))
                elif f == "v_swap_x3":
                    for q_ in range(3):
                        put(m_, ("ins", "v_swap_b32", (self.vT[0][(m_ + q_) % 16], self.vT[0][(m_ + 5 + q_) % 16]), {}))
                elif f == "v_swap_dep":
                    put(m_, ("ins", "v_swap_b32", (self.vT[0][m_ % 16], self.vT[0][(m_ + 3) % 16]), {}))
                    put(m_, ("ins", "v_swap_b32", (self.vT[0][(m_ + 3) % 16], self.vT[0][(m_ + 7) % 16]), {}))
                elif f in ("v_mov_b32",):
                    put(m_, ("ins", f, (self.vF[0], self.vF[1]), {}))
                elif f == "v_add_u32":
                    put(m_, ("ins", f, (self.vF[0], self.s_t[5], self.vF[1]), {}))
                elif f == "v_accvgpr_read_b32":
                    put(m_, ("ins", f, (self.vF[0], self.run[0][m_ % 16] if c.exact else self.acc[0][0]), {}))
                elif f == "v_accvgpr_write_b32":
                    put(m_, ("ins", f, (self.run[0][m_ % 16], self.vF[0]), {}))
                elif f == "v_accvgpr_rw":
                    put(m_, ("ins", "v_accvgpr_read_b32", (self.vF[0], self.run[0][m_ % 16]), {}))
                    put(m_, ("ins", "v_accvgpr_write_b32", (self.run[1][m_ % 16], self.vF[1]), {}))
                elif f == "v_add_f32":
                    put(m_, ("ins", f, (self.vF[0], self.vF[1], self.vF[2]), {}))
                elif f == "v_mov_x2":
                    put(m_, ("ins", "v_mov_b32", (self.vF[0], self.vF[1]), {}))
                    put(m_, ("ins", "v_mov_b32", (self.vF[2], self.vF[3]), {}))
                elif f == "v_nop":
                    put(m_, ("ins", f, (), {}))
                elif f == "v_mov_sgpr":
                    put(m_, ("ins", "v_mov_b32", (self.vF[0], self.s_t[5]), {}))
                elif f == "v_pk_add_f32":
                    put(m_, ("ins", f, (self.vF.sub(0, 2), self.vF.sub(0, 2), self.vF.sub(2, 2)), {}))
                elif f == "s_nop":
                    put(m_, ("ins", f, (0,), {}))
                elif f == "s_add_u32":
                    put(m_, ("ins", f, (self.s_t[5], self.s_t[5], 1), {}))
                else:
                    raise ValueError(f)
        # ---- emit ----
        if trans:
            # the new tile's second K-tile (requested a body ago) has landed by now: with the queue empty, the C stores below are
            # older than every load this body issues -- the counted waits of the bodies that follow stay exact
            e("s_waitcnt", vmcnt=0)
            self.vmq.clear()
        # the first group's fragments were requested during the previous tile
        self.lg_wait({("R", 0)})
        for op in gaps[-1]:
            self.run_op(op)
        for m, (g, u, b) in enumerate(order):
            slot = g & 1
            i, n = b // c.TN, b % c.TN
            srcc = self.acc[b]
            first = fold and g == 0 and u == 0
            if first:
                self.fold_before(b)
                srcc = 0
            self.emit_mfma(b, slot, i, n, u, srcc)
            if first:
                if trans:
                    self.trans_after(b)
                else:
                    self.fold_after(b)
            for op in gaps[m]:
                self.run_op(op)
        assert len(order) == c.NMF
        self.stA, self.stB = self.st_sets[0]

    # ------------------------------------------------------------------ main loop + epilogue
    def main_loop(self):
        c, p = self.c, self.p
        e = p.emit
        state0 = (list(self.vmq), list(self.lgq))
        if True:
            # One K-tile body per LDS stage (and per kind: ordinary / first tile of a kc slice): a body names its stage's
            # address registers, so the loop computes no address -- any VALU op beside the MFMA stream costs matrix-pipe
            # time; rotating register triples with v_swap_b32 instead of unrolling measured 4 % slower
            # (profiles/r03/asm_probe_v6.jsonl).  Every body ends with the same dispatch: done? -> slice boundary? -> the
            # next stage's body.
            L_done = p.label("done")
            # deep: body j of six multiplies a tile in LDS stage j % 3 and stages with register set (j + 1) & 1
            NBODY = 6 if c.deep else 3
            ahead = 3 if c.deep else 2            # a body requests the tile `ahead` tiles past the one it multiplies
            N = [p.label(f"tile_s{k}") for k in range(NBODY)]
            F = [p.label(f"fold_s{k}") for k in range(NBODY)] if c.exact else None
            e("s_mov_b32", self.s_cnt, c.KC_TILES if c.exact else 0x7fffffff)

            def rset(j):
                return ((j + 1) & 1) if c.deep else 0

            L_fin = [p.label(f"fin_s{k}") for k in range(NBODY)] if c.pipe else None
            TR = [p.label(f"trans_s{k}") for k in range(NBODY)] if c.pipe else None
            sw_stub, sw_ret = {}, {}

            def tail(k, fall_through, site=None):
                nk = (k + 1) % NBODY
                e("s_sub_u32", self.s_rem, self.s_rem, 1)
                e("s_cmp_eq_u32", self.s_rem, 0)
                e("s_cbranch_scc1", L_fin[nk] if c.pipe else L_done)
                if c.pipe:
                    # two K-tiles left: every load of this tile is on its way -- the next tile's can follow (pipe_switch)
                    sw_stub[site], sw_ret[site] = p.label(f"swstub{site}"), p.label(f"swret{site}")
                    e("s_cmp_eq_u32", self.s_rem, 2)
                    e("s_cbranch_scc1", sw_stub[site])
                    p.place(sw_ret[site])
                self.tail_mask_if(self.s_rem, ahead + 1)    # the next body loads the last K-tile
                self.stA, self.stB = self.st_sets[rset(nk)]
                self.mask_last_pieces_if(self.s_rem, 2)   # the next body stores it (K % 4 != 0: zero what lies beyond K)
                self.stA, self.stB = self.st_sets[0]
                if c.exact:
                    e("s_sub_u32", self.s_cnt, self.s_cnt, 1)
                    e("s_cmp_eq_u32", self.s_cnt, 0)
                    e("s_cbranch_scc1", F[nk])
                if not fall_through:
                    e("s_branch", N[nk])
            e("raw", ".p2align 6")
            for k in range(NBODY):
                p.place(N[k])
                self.tile_body(False, stage=k % 3, regset=rset(k))
                assert (self.vmq, self.lgq) == state0, "loop-carried queue state differs"
                tail(k, fall_through=(k < NBODY - 1), site=k)
            if c.exact:
                assert c.KC_TILES > 1
                for k in range(NBODY):
                    p.place(F[k])
                    e("s_mov_b32", self.s_cnt, c.KC_TILES)
                    self.tile_body(True, stage=k % 3, regset=rset(k))
                    assert (self.vmq, self.lgq) == state0, "loop-carried queue state differs (fold tile)"
                    tail(k, fall_through=False, site=NBODY + k)
            if c.pipe:
                # the tile is done.  Armed (pipe_switch pointed the last two bodies' loads at the next tile): straight on into that
                # tile's transition body -- LDS stage k holds its first K-tile, the staging registers wait for its second
                for k in range(NBODY):
                    L_go = p.label(f"go_s{k}")
                    p.place(L_fin[k])
                    e("s_bitcmp1_b32", self.s_pipe, 1)
                    e("s_cbranch_scc0", L_done)
                    e("s_andn2_b32", self.s_pipe, self.s_pipe, 2)
                    e("s_add_u32", self.s_rem, self.s_K, c.BK - 1)
                    e("s_lshr_b32", self.s_rem, self.s_rem, (c.BK).bit_length() - 1)
                    e("s_branch", TR[k])
                for k in range(NBODY):
                    p.place(TR[k])
                    if c.exact:
                        e("s_mov_b32", self.s_cnt, c.KC_TILES)
                    self.tile_body(True, stage=k % 3, regset=rset(k), trans=True)
                    assert ([t_ for t_ in self.vmq if t_[0] != "S"], self.lgq) == state0, "loop-carried queue state differs (transition tile)"
                    self.vmq = list(state0[0])      # (the C stores are older than every load in flight: the next bodies' counted waits cover them)
                    tail(k, fall_through=False, site=2 * NBODY + k)
                self.outlined_blocks.append((p.label("swblock"), lambda: self.pipe_switch(sw_stub, sw_ret)))
            p.place(L_done)
            return

    def c_addr_setup(self):
        """vC[n] = byte offset in C of this lane's first element of block column n (row m0 + wm0 + 4*hi, col n0 + wn0 + lo + 32n);
        columns beyond N get an offset the bounds check always rejects (C is kept under 2 GB by the launcher)"""
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        lane, lo, hi = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, v(0))
        e("v_and_b32", lo, 31, lane)
        e("v_lshrrev_b32", hi, 5, lane)
        # row = m0 + wm0 + 4 * hi (+ 32 i + 8 q + rr); col = n0 + wn0 + lo (+ 32 n)
        e("s_add_u32", st[0], self.s_m0, self.s_wm0)
        e("v_lshl_add_u32", t[3], hi, 2, st[0])
        e("v_mul_lo_u32", t[3], t[3], self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], lo)              # col of block n = 0
        dense = getattr(self, "s_csC4", None) is None      # (convolution / integer kernels: C's columns are adjacent)
        if dense:
            e("v_lshl_add_u32", t[3], t[4], 2, t[3])     # byte offset of (row, col)
        else:
            e("v_mul_lo_u32", t[6], t[4], self.s_csC4)
            e("v_add_u32", t[3], t[3], t[6])             # byte offset of (row, col): row * ldc * 4 + col * csC * 4
            e("s_lshl_b32", st[5], self.s_csC4, 5)       # 32 columns further
        for n in range(c.TN):
            e("v_add_u32", t[5], 32 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            if dense:
                e("v_add_u32", t[6], 128 * n, t[3])
            elif n == 0:
                e("v_mov_b32", t[6], t[3])
            else:
                e("v_add_u32", t[6], st[5], t[6])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[6], VCC)
            self.dump(f"vC[{n}]", self.vC[n])

    def c_step(self, i, q, rr):
        """vC moves on to the next accumulator row: + ldc (rows rr within a quad), + 5*ldc - 0 (next quad: rows 8 apart)"""
        c = self.c
        if i == c.TM - 1 and q == 3 and rr == 3:
            return
        step = self.s_ldc4 if rr < 3 else self.s_ldc20
        for n in range(c.TN):
            self.p.emit("v_add_u32", self.vC[n], step, self.vC[n])

    def load_beta_c(self):
        """laser-order kernels, beta != 0: the running sum starts as beta * C0 (one rounding; gemm_ukernel_generic.nim:59-66,
        gemm.nim:158 -- beta applies with the first kc slice only), before the first slice is added.  Loads go through the
        fragment registers (free until the first fragment read) and are drained inside this block, so the counted waits of the
        tile loads (computed for the beta == 0 path) stay correct."""
        c, p = self.c, self.p
        e = p.emit
        skip = p.label("nobeta")
        e("s_and_b32", self.s_t[0], self.s_beta, 0x7fffffff)
        e("s_cmp_eq_u32", self.s_t[0], 0)
        e("s_cbranch_scc1", skip)
        self.c_addr_setup()
        pool = [r[k] for slot in range(2) for r in (self.fa[slot] + self.fb[slot]) for k in range(4)]
        per = 4 * c.TN
        assert len(pool) >= per
        for i in range(c.TM):
            for q in range(4):
                for rr in range(4):
                    for n in range(c.TN):
                        # (runv: straight into the running sum's own registers -- the fragment registers are AGPRs there)
                        dst = self.run[i * c.TN + n][4 * q + rr] if c.runv else pool[rr * c.TN + n]
                        e("buffer_load_dword", dst, self.vC[n], self.srdC, 0, offen=True)
                    self.c_step(i, q, rr)
                e("s_waitcnt", vmcnt=0)
                for rr in range(4):
                    for n in range(c.TN):
                        if c.runv:
                            x = self.run[i * c.TN + n][4 * q + rr]
                            e("v_mul_f32", x, self.s_beta, x)
                            continue
                        x = pool[rr * c.TN + n]
                        e("v_mul_f32", x, self.s_beta, x)
                        e("v_accvgpr_write_b32", self.run[i * c.TN + n][4 * q + rr], x)
        p.place(skip)

    def fused_epilogue(self):
        """bias pointer != 0 or activation != 0: C = act((run | 0) + alpha * sum + bias), the bias added with one more rounding after
        the last slice, relu = max(x, 0) (the reference plans this fusion: README.md:238-242, TODOs gemm.nim:196); a path of its own so
        that the plain epilogue carries none of it.  One-chain kernels: beta == 0 only (the launcher guarantees it).  Ends the program."""
        c, p = self.c, self.p
        e, t, st = p.emit, self.vt, self.s_t
        plain = p.label("plain")
        e("s_load_dwordx2", self.srdBias.sub(0, 2), s(0, 2), KA_BIAS)
        e("s_load_dwordx4", self.s_epi, s(0, 2), KA_EPI)
        e("s_waitcnt", lgkmcnt=0)
        e("s_or_b32", st[0], self.srdBias[0], self.srdBias[1])
        e("s_or_b32", st[1], st[0], self.s_epi[2])
        e("s_cmp_eq_u32", st[1], 0)
        e("s_cbranch_scc1", plain)
        # bias descriptor: a null pointer gets a zero-size buffer (every load returns 0: x + 0 is x)
        e("s_and_b32", self.srdBias[1], self.srdBias[1], 0xffff)
        # bytes of the bias view: ((M - 1) * rowStride + (N - 1) * colStride + 1) * 4 -- rows / columns of a ragged tile beyond it read 0
        e("s_sub_u32", st[2], self.s_M, 1)
        e("s_mul_i32", st[2], st[2], self.s_epi[0])
        e("s_sub_u32", st[3], self.s_N, 1)
        e("s_mul_i32", st[3], st[3], self.s_epi[1])
        e("s_add_u32", st[2], st[2], st[3])
        e("s_add_u32", st[2], st[2], 1)
        e("s_lshl_b32", st[2], st[2], 2)
        e("s_cmp_eq_u32", st[0], 0)
        e("s_cselect_b32", self.srdBias[2], 0, st[2])
        e("s_mov_b32", self.srdBias[3], 0x00020000)
        e("s_lshl_b32", st[2], self.s_epi[0], 2)          # rowStrideBias * 4
        e("s_mul_i32", st[3], st[2], 5)
        self.c_addr_setup()
        # bias offsets of this lane's first element of block column n: (row * rsBias + col * csBias) * 4, rows / cols as in c_addr_setup
        lane, lo, hi = t[0], t[1], t[2]
        vB = [self.vT[0][12 + n] if n < 4 else None for n in range(c.TN)]
        e("s_add_u32", st[0], self.s_m0, self.s_wm0)
        e("v_lshl_add_u32", t[3], hi, 2, st[0])
        e("v_mul_lo_u32", t[3], t[3], st[2])
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], lo)
        e("s_lshl_b32", st[4], self.s_epi[1], 2)          # colStrideBias * 4
        e("v_mul_lo_u32", t[4], t[4], st[4])
        e("v_add_u32", t[3], t[3], t[4])
        e("s_lshl_b32", st[4], st[4], 5)                  # 32 columns further
        for n in range(c.TN):
            if n == 0:
                e("v_mov_b32", vB[0], t[3])
            else:
                e("v_add_u32", vB[n], st[4], vB[n - 1])
        P = self.vT[0]
        norelu = None
        for i in range(c.TM):
            for q in range(4):
                # this lane's 4 rows x TN block columns of the quad: bias loads first (one wait), then the arithmetic
                per = c.TN
                for rr in range(4):
                    for n in range(c.TN):
                        if rr * per + n < 12:
                            e("buffer_load_dword", P[rr * per + n], vB[n], self.srdBias, 0, offen=True)
                    if rr < 3:
                        for n in range(c.TN):
                            e("v_add_u32", vB[n], st[2], vB[n])
                assert 4 * per <= 12 or c.TN == 4
                if c.TN == 4:      # 16 values per quad do not fit beside the offsets: the last row in a second batch
                    pass
                e("s_waitcnt", vmcnt=0)
                for rr in range(4):
                    r = 4 * q + rr
                    if c.TN == 4 and rr == 3:
                        for n in range(c.TN):
                            e("buffer_load_dword", P[n], vB[n], self.srdBias, 0, offen=True)
                        e("s_waitcnt", vmcnt=0)
                    for n in range(c.TN):
                        b = i * c.TN + n
                        tt, uu = t[(2 * n) % 8], t[(2 * n + 1) % 8]
                        bias = P[n] if (c.TN == 4 and rr == 3) else P[rr * per + n]
                        e("v_accvgpr_read_b32", tt, self.acc[b][r])
                        e("v_mul_f32", tt, self.s_alpha, tt)
                        if c.runv:
                            e("v_add_f32", tt, self.run[b][r], tt)
                        elif c.exact:
                            e("v_accvgpr_read_b32", uu, self.run[b][r])
                            e("v_add_f32", tt, uu, tt)
                        e("v_add_f32", tt, bias, tt)
                    skip = p.label("norelu")
                    e("s_cmp_lg_u32", self.s_epi[2], 1)
                    e("s_cbranch_scc1", skip)
                    for n in range(c.TN):
                        e("v_max_f32", t[(2 * n) % 8], 0, t[(2 * n) % 8])
                    p.place(skip)
                    for n in range(c.TN):
                        e("buffer_store_dword", t[(2 * n) % 8], self.vC[n], self.srdC, 0, offen=True)
                    self.c_step(i, q, rr)
                # bias offsets: on to the next quad (rows 8 apart: + 5 rows from the last row of this one)
                if not (i == c.TM - 1 and q == 3):
                    for n in range(c.TN):
                        e("v_add_u32", vB[n], st[3], vB[n])
        self.end_run()
        p.place(plain)

    def epilogue(self):
        """C = (beta * C0 + alpha * slice sums in order) -- gemm_ukernel_generic.nim:53-76, predicated by the descriptor's bounds check"""
        c, p = self.c, self.p
        e = p.emit
        t = self.vt
        st = self.s_t
        e("s_nop", 15)
        e("s_nop", 7)
        # drain what the loop left in flight (prefetched tile beyond K: never used)
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        self.vmq.clear()
        self.lgq.clear()
        if c.debug:
            for k_ in range(4):
                self.dump(f"acc[0][{k_}]", self.acc[0][k_])
            self.dump("acc[last][15]", self.acc[-1][15])
            if c.exact:
                self.dump("run[0][0]", self.run[0][0])
            for k_ in range(4):
                self.dump(f"srdC[{k_}]", self.srdC[k_])
            self.dump("s_rem", self.s_rem)
        self.mode_dispatch()
        self.fused_epilogue()
        self.c_addr_setup()
        if c.exact:
            # C = run + alpha * (the last slice's sum); run already carries beta * C0 and the earlier slices
            def rows(i, q):
                for rr in range(4):
                    r = 4 * q + rr
                    for n in range(c.TN):
                        b = i * c.TN + n
                        tt, uu = t[(2 * n) % 8], t[(2 * n + 1) % 8]
                        e("v_accvgpr_read_b32", tt, self.acc[b][r])
                        e("v_mul_f32", tt, self.s_alpha, tt)
                        if c.runv:
                            e("v_add_f32", tt, self.run[b][r], tt)
                        else:
                            e("v_accvgpr_read_b32", uu, self.run[b][r])
                            e("v_add_f32", tt, uu, tt)
                        if "cstores" not in c.ablate:
                            e("buffer_store_dword", tt, self.vC[n], self.srdC, 0, offen=True)
                    self.c_step(i, q, rr)
            for i in range(c.TM):
                for q in range(4):
                    rows(i, q)
        else:
            # one chain: C = beta * C0 + alpha * sum (beta == 0: C0 is never read, gemm_ukernel_generic.nim:59-66)
            withc, done = p.label("beta"), p.label("stored")
            e("s_and_b32", st[0], self.s_beta, 0x7fffffff)
            e("s_cmp_lg_u32", st[0], 0)
            e("s_cbranch_scc1", withc)
            for i in range(c.TM):
                for q in range(4):
                    for rr in range(4):
                        r = 4 * q + rr
                        for n in range(c.TN):
                            tt = t[(2 * n) % 8]
                            e("v_accvgpr_read_b32", tt, self.acc[i * c.TN + n][r])
                            e("v_mul_f32", tt, self.s_alpha, tt)
                            if "cstores" not in c.ablate:
                                e("buffer_store_dword", tt, self.vC[n], self.srdC, 0, offen=True)
                        self.c_step(i, q, rr)
            e("s_branch", done)
            p.place(withc)
            self.c_addr_setup()
            P = self.vT[0]
            for i in range(c.TM):
                for q in range(4):
                    for n in range(c.TN):
                        e("v_mov_b32", t[n], self.vC[n])
                    for rr in range(4):
                        for n in range(c.TN):
                            e("buffer_load_dword", P[rr * c.TN + n], self.vC[n], self.srdC, 0, offen=True)
                        if rr < 3:
                            for n in range(c.TN):
                                e("v_add_u32", self.vC[n], self.s_ldc4, self.vC[n])
                    for n in range(c.TN):
                        e("v_mov_b32", self.vC[n], t[n])
                    e("s_waitcnt", vmcnt=0)
                    for rr in range(4):
                        r = 4 * q + rr
                        for n in range(c.TN):
                            tt, x = t[4 + n % 4], P[rr * c.TN + n]
                            e("v_mul_f32", x, self.s_beta, x)
                            e("v_accvgpr_read_b32", tt, self.acc[i * c.TN + n][r])
                            e("v_mul_f32", tt, self.s_alpha, tt)
                            e("v_add_f32", tt, x, tt)
                            e("buffer_store_dword", tt, self.vC[n], self.srdC, 0, offen=True)
                        self.c_step(i, q, rr)
            p.place(done)
        self.end_run()

    # ------------------------------------------------------------------ runs that share a tile: workspace, flags, ordered fix-up
    def end_run(self):
        if self.c.persistent or self.c.cpers:
            self.p.emit("s_branch", self.L_run)
        else:
            self.p.emit("s_endpgm")

    def tile_bytes(self):
        c = self.c
        return c.NB * c.ACCR * 256 * 4

    def ws_descriptors(self, slot):
        """srdA = flags[slot] (one dword: of the workgroup's 256 byte offsets 4 * tid only thread 0's is in range), srdB = workspace
        slot `slot`: one tile in register order -- dword (b * ACCR + r) * 256 + tid = register r of block b of thread tid.
        Clobbers s_t[4], s_t[5]; `slot` must not be one of them."""
        e, st = self.p.emit, self.s_t
        e("s_load_dwordx4", self.srdB, s(0, 2), KA_WS)
        e("s_waitcnt", lgkmcnt=0)
        e("s_lshl_b32", st[5], slot, 2)
        e("s_add_u32", self.srdA[0], self.srdB[2], st[5])
        e("s_addc_u32", self.srdA[1], self.srdB[3], 0)
        e("s_and_b32", self.srdA[1], self.srdA[1], 0xffff)
        e("s_mov_b32", self.srdA[2], 4)
        e("s_mov_b32", self.srdA[3], 0x00020000)
        e("s_mul_hi_u32", st[4], slot, self.tile_bytes())
        e("s_mul_i32", st[5], slot, self.tile_bytes())
        e("s_add_u32", self.srdB[0], self.srdB[0], st[5])
        e("s_addc_u32", self.srdB[1], self.srdB[1], st[4])
        e("s_and_b32", self.srdB[1], self.srdB[1], 0xffff)
        e("s_mov_b32", self.srdB[2], self.tile_bytes())
        e("s_mov_b32", self.srdB[3], 0x00020000)

    def ws_walk(self, fn):
        """fn(b, r, soffset SGPR, immediate offset) for every accumulator register in workspace order; soffset = s_t[5]"""
        c, e, st = self.c, self.p.emit, self.s_t
        e("s_mov_b32", st[5], 0)
        n = 0
        for b in range(c.NB):
            for r in range(c.ACCR):
                fn(b, r, st[5], (n % 4) * 1024)
                n += 1
                if n % 4 == 0 and n < c.NB * c.ACCR:
                    e("s_add_u32", st[5], st[5], 4096)

    def send_block(self):
        """SEND: the running sum of this tile's first slices (beta * C0 included; one chain: the partial chain sum) goes to workspace slot
        `vid`, then flag `vid` is set; workgroup vid + 1 continues the tile from it.
        Visibility across XCDs (each has its own L2): every access to the workspace and the flags carries sc1 = agent scope -- the
        stores write through, the loads do not hit a stale line -- and the flag is stored after the sum's stores have completed
        (s_waitcnt vmcnt(0) on every wave, then the barrier).  The whole-L2 write-back + invalidate pair hipcc emits around a
        release / acquire (buffer_wbl2 sc1 / buffer_inv sc1) protects accesses WITHOUT scope bits; here it only threw the other
        workgroups' operand panels out of the L2: cut launches ran at half speed with it (profiles/r04/plan_sweep_f32_mid_a.jsonl)."""
        c, e, t = self.c, self.p.emit, self.vt
        if c.exact:
            self.fold_all()
        self.ws_descriptors(self.s_vid)
        e("v_lshlrev_b32", t[8], 2, v(0))
        src = self.run if c.exact else self.acc
        self.ws_walk(lambda b, r, so, imm: e("buffer_store_dword", src[b][r], t[8], self.srdB, so, offen=True, offset=imm, sc1=True))
        e("s_waitcnt", vmcnt=0)
        e("s_barrier")
        e("v_mov_b32", t[9], 1)
        e("buffer_store_dword", t[9], t[8], self.srdA, 0, offen=True, sc1=True)
        e("s_waitcnt", vmcnt=0)
        e("s_branch", self.L_run)

    def fold_all(self):
        """run += alpha * acc for every block (the slice fold of the K loop, outside it)"""
        for b in range(self.c.NB):
            self.fold_block(b)

    def fold_block(self, b):
        e, T = self.p.emit, self.vT[0]
        for r in range(16):
            e("v_accvgpr_read_b32", T[r], self.acc[b][r])
        for r in range(16):
            e("v_mul_f32", T[r], self.s_alpha, T[r])      # (1.0 * x is x: no branch here, the hand-over is not the hot loop)
        if self.c.runv:
            for j in range(8):
                e("v_pk_add_f32", self.run[b].sub(2 * j, 2), self.run[b].sub(2 * j, 2), T.sub(2 * j, 2))
            return
        for r in range(16):
            tt = self.vt[r % 4]
            e("v_accvgpr_read_b32", tt, self.run[b][r])
            e("v_add_f32", tt, tt, T[r])
            e("v_accvgpr_write_b32", self.run[b][r], tt)

    def acc_add_block(self, b):
        """acc[b] += the partial in vT"""
        e, T = self.p.emit, self.vT[0]
        for r in range(16):
            tt = self.vt[r % 4]
            e("v_accvgpr_read_b32", tt, self.acc[b][r])
            e("v_add_f32", tt, tt, T[r])
            e("v_accvgpr_write_b32", self.acc[b][r], tt)

    def load_received(self, t_off):
        """laser-order: the received sum becomes the running sum; one chain: acc += received partial"""
        c, e, T, st = self.c, self.p.emit, self.vT[0], self.s_t
        if c.exact:
            self.ws_walk(lambda b, r, so, imm: e("buffer_load_dword", self.run[b][r], t_off, self.srdB, so, offen=True, offset=imm, sc1=True))
            e("s_waitcnt", vmcnt=0)
            return
        e("s_mov_b32", st[5], 0)
        n = 0
        for b in range(c.NB):
            for r in range(c.ACCR):
                e("buffer_load_dword", T[r], t_off, self.srdB, st[5], offen=True, offset=(n % 4) * 1024, sc1=True)
                n += 1
                if n % 4 == 0:
                    e("s_add_u32", st[5], st[5], 4096)
            e("s_waitcnt", vmcnt=0)
            self.acc_add_block(b)

    def recv_block(self, L_epi, L_send):
        """RECEIVE: wait for flag vid - 1 (set by the previous workgroup after the first thing it did), take the sum from slot vid - 1,
        clear the flag for the next launch; then: the run set-up (the sum arrived before the piece started: phase 5), or run B on top
        of it (laser-order, more slices left), or the end of the tile (store C / send the sum on)."""
        c, p = self.c, self.p
        e, t, st, sc = p.emit, self.vt, self.s_t, self.s_sc
        slot, fv = st[3], st[2]
        spin, got, lost, after = p.label("spin"), p.label("got"), p.label("lost"), p.label("after")
        e("s_sub_u32", slot, self.s_vid, 1)
        self.ws_descriptors(slot)
        e("v_lshlrev_b32", t[8], 2, v(0))
        # flags bit 3 (tests only): the receiver gives up at once, as if its ~2 s of polling had run out -- the error report below and
        # the launcher's handling of it are then exercised without a hung sender
        e("s_load_dword", st[0], s(0, 2), KA_SCHED2 + 24)
        e("s_waitcnt", lgkmcnt=0)
        e("s_bitcmp1_b32", st[0], 3)
        e("s_cbranch_scc1", lost)
        e("s_mov_b32", st[5], 0)
        p.place(spin)
        e("buffer_load_dword", t[9], OFF, self.srdA, 0, sc1=True)
        e("s_waitcnt", vmcnt=0)
        e("v_readfirstlane_b32", fv, t[9])
        e("s_cmp_lg_u32", fv, 0)
        e("s_cbranch_scc1", got)
        # a legitimate wait is short (the sum is the first thing its sender computes); after ~2 s of polling the workgroup COUNTS itself
        # in the error word in front of the flags (an atomic add: every workgroup that gave up is in the sum) and goes on -- a wrong
        # result that is flagged, not a hung GPU.  The launcher reads the word back behind every cut launch and fails the next call
        # on the stream (gemm_f32_asm.cpp check_stream_poison): the reference aborts on a violated precondition
        # (gemm_prepacked.nim:125); this library never returns a wrong C silently.
        e("s_add_u32", st[5], st[5], 1)
        e("s_cmp_lt_u32", st[5], 1 << 21)
        e("s_cbranch_scc0", lost)
        e("s_sleep", 8)
        e("s_branch", spin)
        p.place(lost)
        e("s_load_dwordx2", self.s_sc.sub(0, 2), s(0, 2), KA_WS + 8)
        e("s_waitcnt", lgkmcnt=0)
        srdE = self.s_sc.sub(0, 4)          # (the scheduler constants are reloaded where they are next used; srdC stays what the epilogue needs)
        e("s_sub_u32", srdE[0], sc[0], 4)
        e("s_subb_u32", srdE[1], sc[1], 0)
        e("s_and_b32", srdE[1], srdE[1], 0xffff)
        e("s_mov_b32", srdE[2], 4)
        e("s_mov_b32", srdE[3], 0x00020000)
        e("v_mov_b32", t[9], 1)
        e("buffer_atomic_add", t[9], t[8], srdE, 0, offen=True, sc1=True)     # (4 * tid: only thread 0 of the workgroup is in range)
        e("s_waitcnt", vmcnt=0)
        p.place(got)
        self.load_received(t[8])
        e("s_barrier", comment="every wave has seen the flag: it can be cleared for the next launch")
        e("v_mov_b32", t[9], 0)
        e("buffer_store_dword", t[9], t[8], self.srdA, 0, offen=True, sc1=True)
        e("s_waitcnt", vmcnt=0)
        if c.exact:
            e("s_cmp_lg_u32", self.s_phase, 5)
            e("s_cbranch_scc1", after)
            e("s_mov_b32", self.s_phase, 4)
            e("s_branch", self.L_setup)
            p.place(after)
            # after run A: slices p0 + 1 .. pz - 1 left?
            last = p.label("lastslice")
            e("s_add_u32", st[0], self.s_p0, 1)
            e("s_cmp_ge_u32", st[0], self.s_pz)
            e("s_cbranch_scc1", last)
            self.fold_all()
            e("s_load_dwordx8", sc, s(0, 2), KA_SCHED2)
            e("s_waitcnt", lgkmcnt=0)
            e("s_add_u32", st[0], self.s_p0, 1)
            e("s_mul_i32", self.s_kb, st[0], sc[5])
            e("s_mul_i32", st[0], self.s_pz, sc[5])
            e("s_min_u32", st[0], st[0], self.s_K)
            e("s_sub_u32", self.s_Keff, st[0], self.s_kb)
            e("s_mov_b32", self.s_mode, MODE_CONT)
            e("s_mov_b32", self.s_end, self.s_fin)
            e("s_branch", self.L_setup)
            p.place(last)
        e("s_cmp_eq_u32", self.s_fin, END_EPI)
        e("s_cbranch_scc1", L_epi)
        e("s_branch", L_send)

    def mode_dispatch(self):
        """persistent kernels, after the K loop: store C (the ordinary epilogue), send the running sum on, or receive"""
        c, p = self.c, self.p
        if not c.persistent:
            return
        e = p.emit
        L_send, L_epi = p.label("send"), p.label("epi")
        e("s_cmp_eq_u32", self.s_end, END_SEND)
        e("s_cbranch_scc1", L_send)
        e("s_cmp_eq_u32", self.s_end, END_RECV)
        e("s_cbranch_scc1", self.L_recv)
        p.place(L_epi)
        self.outlined_blocks.append((L_send, self.send_block))
        self.outlined_blocks.append((self.L_recv, lambda: self.recv_block(L_epi, L_send)))

    def build(self):
        self.outlined = []
        self.outlined_blocks = []
        self.prologue()
        self.main_loop()
        self.epilogue()
        for label, ins, back in self.outlined:
            self.p.place(label)
            for i in ins:
                self.p.emit(*i)
            self.p.emit("s_branch", back)
        for label, fn in self.outlined_blocks:
            self.p.place(label)
            fn()
        if self.c.persistent or self.c.cpers:
            self.p.place(self.L_exit)
            self.p.emit("s_endpgm")
        return self.p


def make(name, **over):
    kw = dict(CONFIGS[name])
    kw.update(over)
    return Gen(Cfg(name, **kw))


def kernel_text(gen, symbol):
    """complete .s file: code + kernel descriptor + code-object metadata"""
    c = gen.c
    body = gen.p.text()
    nv = (gen.p._v + 7) // 8 * 8          # arch VGPRs (allocation granule 8); the AGPRs follow at accum_offset
    na = (gen.p._a + 7) // 8 * 8
    return f"""\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.text
\t.globl\t{symbol}
\t.p2align\t8
\t.type\t{symbol},@function
{symbol}:
{body}.Lfunc_end_{symbol}:
\t.size\t{symbol}, .Lfunc_end_{symbol}-{symbol}
\t.rodata
\t.p2align\t6
\t.amdhsa_kernel {symbol}
\t\t.amdhsa_group_segment_fixed_size {c.lds_alloc}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {KERNARG_SIZE}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr {nv + na}
\t\t.amdhsa_next_free_sgpr 100
\t\t.amdhsa_accum_offset {nv}
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.amdgpu_metadata
---
amdhsa.version: [1, 2]
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.kernels:
  - .name: {symbol}
    .symbol: {symbol}.kd
    .kernarg_segment_size: {KERNARG_SIZE}
    .kernarg_segment_align: 8
    .group_segment_fixed_size: {c.lds_alloc}
    .private_segment_fixed_size: 0
    .wavefront_size: 64
    .sgpr_count: 100
    .vgpr_count: {nv}
    .agpr_count: {na}
    .max_flat_workgroup_size: 256
    .args:
      - {{.size: 8, .offset: 0, .value_kind: global_buffer, .address_space: global}}
      - {{.size: 8, .offset: 8, .value_kind: global_buffer, .address_space: global}}
      - {{.size: 8, .offset: 16, .value_kind: global_buffer, .address_space: global}}
      - {{.size: 8, .offset: 24, .value_kind: global_buffer, .address_space: global}}
      - {{.size: 4, .offset: 32, .value_kind: by_value}}
      - {{.size: 4, .offset: 36, .value_kind: by_value}}
      - {{.size: 4, .offset: 40, .value_kind: by_value}}
      - {{.size: 4, .offset: 44, .value_kind: by_value}}
      - {{.size: 4, .offset: 48, .value_kind: by_value}}
      - {{.size: 4, .offset: 52, .value_kind: by_value}}
      - {{.size: 8, .offset: 56, .value_kind: by_value}}
      - {{.size: 8, .offset: 64, .value_kind: global_buffer, .address_space: global}}
      - {{.size: {KERNARG_SIZE - 72}, .offset: 72, .value_kind: by_value}}
...
\t.end_amdgpu_metadata
"""


if __name__ == "__main__":
    import argparse
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for name in CONFIGS:
        g = make(name)
        g.build()
        sym = "lh_f32_" + name
        with open(os.path.join(args.out, sym + ".s"), "w") as f:
            f.write(kernel_text(g, sym))
        print(sym, len(g.p.ins), "instructions")
