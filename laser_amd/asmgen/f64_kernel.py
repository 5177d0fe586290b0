"""float64 GEMM kernels for gfx950, hand-scheduled: the f32 generator's program structure (f32_kernel.Gen: 3-stage LDS ring,
one barrier per K-tile, one body per stage, counted waits, accumulators in AGPRs, laser-order fold every kc) with the f64
matrix instruction and its own LDS image.

  v_mfma_f64_16x16x4_f64   16x16 block, 4 k per instruction (64 cycles per SIMD on MI355X), lane l feeds A[l % 16][l / 16] and
                           B[l / 16][l % 16] and holds D[l / 16 + 4 * d][l % 16], d = 0..3: 8 accumulator registers per block
  kc = 256                 gemm_tiling.nim:310 (2048 bytes / sizeof(float64)): the running sum is folded every 16 K-tiles of 16

LDS image (per stage: A panel rows 0..BM-1, then the B panel rows = columns of the tile): a row holds BK = 16 doubles as 8
chunks of 16 bytes, chunk 4*(k / 8) + k % 4 holding k and k + 4 -- the two k-steps lane group q = k % 4 needs from one
ds_read_b128 -- and is 144 bytes long: with 9 chunks per row the 16 rows a 16-lane group reads fall into 16 different
4-bank groups, so the fragment reads are conflict-free without a swizzle, and every store is ONE ds_write2_b64:
  A piece (16 bytes = k0, k0 + 1 of one row)      -> chunks c and c + 1 of that row:           offsets 0, +16 bytes
  B piece (16 bytes = columns x0, x0 + 1 of one k) -> the same chunk of rows x0 and x0 + 1:     offsets 0, +144 bytes
Operands: A row-major (k-contiguous), B row-major (x-contiguous) or passed transposed (`_nt`: k-contiguous, stored like A), C
row-major; K a multiple of 2; any alpha / beta (float64 in the kernel arguments)."""
from .core import v, a, s, VCC
from .f32_kernel import Gen, Cfg, kernel_text, KA_A, KA_LDA, KA_DBG, KA_CONV1  # noqa: F401

CONFIGS = {
    # one wave per SIMD: 2 x 2 waves of 64x64 = 16 blocks = 128 accumulator registers (+ 128 for the running sum)
    "exact_128x128x16": dict(BM=128, BN=128, BK=16, exact=True, runv=True, dataa=True, pipe=True),
    "fast_128x128x16": dict(BM=128, BN=128, BK=16, exact=False, pipe=True),
    # problems of few tiles (the reference's f64 bench shape, 960^3 = 225 tiles): 2 x 2 waves of 32x32, several workgroups per CU
    "exact_64x64x16": dict(BM=64, BN=64, BK=16, exact=True, runv=True, pipe=True),
    "fast_64x64x16": dict(BM=64, BN=64, BK=16, exact=False, pipe=True),
    # B passed transposed (rowStrideB == 1: k-contiguous like A)
    "exact_128x128x16_nt": dict(BM=128, BN=128, BK=16, exact=True, b_kcontig=True, runv=True, dataa=True, pipe=True),
    "fast_128x128x16_nt": dict(BM=128, BN=128, BK=16, exact=False, b_kcontig=True, pipe=True),
    "exact_64x64x16_nt": dict(BM=64, BN=64, BK=16, exact=True, b_kcontig=True, runv=True, pipe=True),
    "fast_64x64x16_nt": dict(BM=64, BN=64, BK=16, exact=False, b_kcontig=True, pipe=True),
}
KA_ALPHA64 = 72      # alpha, beta as float64 (the f32 kernels' float fields at 56 / 60 are unused here)
KA_BSA64 = 88        # batch stride of A in bytes (u64); B's and C's at 112 / 120 as in the f32 kernels


class Gen64(Gen):
    # ------------------------------------------------------------------ registers
    def alloc(self):
        c, p = self.c, self.p
        S, V = p.salloc, p.valloc
        self.ka0 = S(8, align=4)
        self.ka1 = S(8, align=4)
        self.s_lda, self.s_ldb, self.s_ldc, self.s_M, self.s_N, self.s_K = (self.ka1[i] for i in range(6))
        self.s_alpha, self.s_beta = self.ka1[6], self.ka1[7]
        self.srdA, self.srdB, self.srdC = S(4), S(4), S(4)
        self.s_rem, self.s_cnt = S(), S()
        self.s_bstep = S()
        self.s_m0, self.s_n0, self.s_wave, self.s_wm0, self.s_wn0 = S(), S(), S(), S(), S()
        self.s_t = [S() for _ in range(6)]
        self.s_ab = S(4, align=4)                 # alpha (2 registers), beta (2 registers): float64
        self.s_al, self.s_be = self.s_ab.sub(0, 2), self.s_ab.sub(2, 2)
        self.s_a1, self.s_b0 = S(), S()           # 0 when alpha == 1.0 / when beta == +-0.0
        self.s_bsA, self.s_bsBC = S(2, align=2), S(4, align=4)   # batch strides in bytes (grid y = batch index)
        self.s_ldc4, self.s_ldc20 = S(), S()      # here: ldc * 8 bytes, 4 * ldc * 8 (the next accumulator row of a lane)
        self.s_csC4 = None
        if c.pipe:
            # pipelined tile transitions (f32_kernel.py Cfg.pipe): bit 0 = this launch may pipeline, bit 1 = armed
            self.s_pipe = S()
            self.srdCd = S(4)        # C from this wave's first row of the tile being finished (rows through the stores' scalar offset)
        self.alloc_sched()
        self.acc = [p.aalloc(8) for _ in range(c.NB)]
        # runv / dataa (f32_kernel.py Cfg, round 6): the running sum in arch VGPRs (the slice fold of a block is 8 v_accvgpr_read + 4
        # v_add_f64 where the all-AGPR plan has 28 VALU operations), fragments + staging in AGPRs where the VGPR file would overflow
        self.run = [V(8) if c.runv else p.aalloc(8) for _ in range(c.NB)] if c.exact else None
        D4 = (lambda n: p.aalloc(n)) if c.dataa else V
        self.fa = [[D4(4) for _ in range(c.TM)] for _ in range(2)]
        self.fb = [[D4(4) for _ in range(c.TN)] for _ in range(2)]
        self.stA = [D4(4) for _ in range(c.NPA)]
        self.stB = [D4(4) for _ in range(c.NPB)]
        self.st_sets = [(self.stA, self.stB)]
        self.RA = [[V() for _ in range(3)] for _ in range(c.NG)]
        self.RB = [[V() for _ in range(3)] for _ in range(c.NG)]
        self.WA = [[V() for _ in range(3)] for _ in range(c.NPA)]   # [piece][stage]
        self.WB = [[V() for _ in range(3)] for _ in range(c.NPB)]
        self.v_oob = V()
        self.s_tm = S(2)
        self.s_ktail = S()
        self.vVA = [V() for _ in range(c.NPA)]
        self.vVB = [V() for _ in range(c.NPB)]
        self.vC = [V() for _ in range(c.TN)]
        if c.debug:
            self.srdD = S(4)
            self.s_dslot = S()
            self.v_dbg = V()
        self.ndump = 0
        self.dump_names = []
        self.vT = [V(16, align=2)]
        blk = V(12, align=4)
        self.vt = [blk[i] for i in range(10)]
        self.vF, self.vFaddr, self.vFoff = blk.sub(4, 4), blk[10], blk[11]

    # ------------------------------------------------------------------ prologue (f32_kernel.Gen.prologue: once, scheduler, run_setup)
    def once(self):
        c, p = self.c, self.p
        e = p.emit
        t, st = self.vt, self.s_t
        RS = c.RS
        p.note(f"f64 {c.name}: {c.BM}x{c.BN}x{c.BK} tile, 4 waves, wave tile {c.WTM}x{c.WTN}, "
               f"{'laser-order (kc = 256 slices)' if c.exact else 'one accumulation chain'}")
        e("s_load_dwordx8", self.ka0, s(0, 2), KA_A)
        e("s_load_dwordx8", self.ka1, s(0, 2), KA_LDA)
        e("s_load_dwordx4", self.s_ab, s(0, 2), KA_ALPHA64)
        # batched problems (gemm_strided_batched; the kc slices of the slice-parallel form): workgroup id y = batch index, operand b at
        # base + b * batch stride (bytes, 64-bit; 0 for plain launches) -- as in the f32 kernels
        e("s_load_dwordx2", self.s_bsA, s(0, 2), KA_BSA64)
        e("s_load_dwordx4", self.s_bsBC, s(0, 2), KA_CONV1 + 8)
        e("s_waitcnt", lgkmcnt=0)
        for ptr, bs in ((self.ka0.sub(0, 2), self.s_bsA), (self.ka0.sub(2, 2), self.s_bsBC.sub(0, 2)), (self.ka0.sub(4, 2), self.s_bsBC.sub(2, 2))):
            e("s_mul_i32", st[2], s(3), bs[0])
            e("s_mul_hi_u32", st[3], s(3), bs[0])
            e("s_mul_i32", st[4], s(3), bs[1])
            e("s_add_u32", st[3], st[3], st[4])
            e("s_add_u32", ptr[0], ptr[0], st[2])
            e("s_addc_u32", ptr[1], ptr[1], st[3])
        e("s_xor_b32", st[2], self.s_al[1], 0x3ff00000)
        e("s_or_b32", self.s_a1, st[2], self.s_al[0])
        e("s_and_b32", st[2], self.s_be[1], 0x7fffffff)
        e("s_or_b32", self.s_b0, st[2], self.s_be[0])
        if c.pipe:
            # tile transitions of this launch may be pipelined: beta == 0 (the next tile's running sum starts at 0), K a multiple of BK
            # (no K-tail masks to undo between tiles), at least three K-tiles (the switch happens two tile bodies before a tile ends)
            e("s_and_b32", st[2], self.s_K, c.BK - 1)
            e("s_or_b32", st[2], st[2], self.s_b0)
            e("s_cmp_eq_u32", st[2], 0)
            e("s_cselect_b32", self.s_pipe, 1, 0)
            e("s_cmp_lt_u32", self.s_K, 3 * c.BK)
            e("s_cselect_b32", self.s_pipe, 0, self.s_pipe)
        if c.debug:
            e("s_load_dwordx2", self.srdD.sub(0, 2), s(0, 2), KA_DBG)
            e("s_waitcnt", lgkmcnt=0)
            e("s_and_b32", self.srdD[1], self.srdD[1], 0xffff)
            e("s_mov_b32", self.srdD[2], 0x10000000)
            e("s_mov_b32", self.srdD[3], 0x00020000)
            e("v_lshlrev_b32", self.v_dbg, 2, v(0))
            self.dump("tid", v(0))
        tid = v(0)
        lane, r16, q = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, tid)
        e("v_lshrrev_b32", t[5], 6, tid)
        e("s_nop", 1, comment="VALU write -> v_readfirstlane of the same VGPR needs wait states")
        e("v_readfirstlane_b32", self.s_wave, t[5])
        e("s_nop", 3)
        e("v_and_b32", r16, 15, lane)
        e("v_lshrrev_b32", q, 4, lane)
        e("s_lshr_b32", st[2], self.s_wave, 1)
        e("s_mul_i32", self.s_wm0, st[2], c.WTM)
        e("s_and_b32", st[2], self.s_wave, 1)
        e("s_mul_i32", self.s_wn0, st[2], c.WTN)
        # fragment reads of group g: (wm0 + r16) * RS [+ BM * RS + (wn0 + r16) * RS for B] + (4g + q) * 16
        e("v_add_u32", t[5], self.s_wm0, r16)
        e("v_mul_u32_u24", t[6], RS, t[5])
        e("v_add_u32", t[5], self.s_wn0, r16)
        e("v_mul_u32_u24", t[7], RS, t[5])
        e("v_add_u32", t[7], c.BM * RS, t[7])
        for g in range(c.NG):
            e("v_lshl_add_u32", t[5], q, 4, 64 * g)
            for R, row in ((self.RA, t[6]), (self.RB, t[7])):
                e("v_add_u32", R[g][0], t[5], row)
                e("v_add_u32", R[g][1], c.STAGE, R[g][0])
                e("v_add_u32", R[g][2], 2 * c.STAGE, R[g][0])
        pc, xr = t[0], t[1]
        e("v_and_b32", pc, 7, tid)
        e("v_lshrrev_b32", xr, 3, tid)
        e("v_mov_b32", self.v_oob, 0x80000000)
        # A pieces: piece column pc = tid % 8 (k0 = 2 pc), row xr = tid / 8 (+ 32 per piece)
        #   LDS: row * RS + (4 * (pc >> 2) + 2 * (pc & 1)) * 16 + ((pc >> 1) & 1) * 8
        e("v_lshrrev_b32", t[5], 2, pc)
        e("v_lshlrev_b32", t[5], 6, t[5])                    # (pc >> 2) * 64
        e("v_and_b32", t[6], 1, pc)
        e("v_lshl_add_u32", t[5], t[6], 5, t[5])             # + (pc & 1) * 32
        e("v_bfe_u32", t[6], pc, 1, 1)
        e("v_lshl_add_u32", t[5], t[6], 3, t[5])             # + ((pc >> 1) & 1) * 8
        e("v_mul_u32_u24", t[6], RS, xr)
        e("v_add_u32", t[5], t[5], t[6])
        for i in range(c.NPA):
            e("v_add_u32", self.WA[i][2], 32 * RS * i, t[5])
            e("v_add_u32", self.WA[i][0], c.STAGE, self.WA[i][2])
            e("v_add_u32", self.WA[i][1], 2 * c.STAGE, self.WA[i][2])
        e("s_lshl_b32", st[5], self.s_ldb, 3, comment="ldb * 8 bytes")
        if c.b_kcontig:
            # B passed transposed: its pieces are (column x, k0 .. k0 + 1) -- the A layout with the B panel's offsets
            e("v_add_u32", t[5], c.BM * RS, t[5])
            for j in range(c.NPB):
                e("v_add_u32", self.WB[j][2], 32 * RS * j, t[5])
                e("v_add_u32", self.WB[j][0], c.STAGE, self.WB[j][2])
                e("v_add_u32", self.WB[j][1], 2 * c.STAGE, self.WB[j][2])
            e("s_mov_b32", self.s_bstep, c.BK * 8)
        else:
            # B pieces: x pair px = tid % (BN / 2), row k = tid / (BN / 2) (+ KS per piece, KS = 512 / BN)
            #   LDS: BM * RS + 2 px * RS + (4 * (k >> 3) + (k & 3)) * 16 + ((k >> 2) & 1) * 8; k = k0 + KS * j with k0 < KS
            HB = c.BN // 2
            KS = 256 // HB
            px, k0 = t[0], t[1]
            e("v_and_b32", px, HB - 1, tid)
            e("v_lshrrev_b32", k0, HB.bit_length() - 1, tid)
            e("v_and_b32", t[5], 3, k0)
            e("v_lshlrev_b32", t[5], 4, t[5])                    # (k0 & 3) * 16
            e("v_bfe_u32", t[6], k0, 2, 1)
            e("v_lshl_add_u32", t[5], t[6], 3, t[5])             # + ((k0 >> 2) & 1) * 8      (k0 < 8)
            e("v_mul_u32_u24", t[6], 2 * RS, px)
            e("v_add_u32", t[5], t[5], t[6])
            e("v_add_u32", t[5], c.BM * RS, t[5])
            for j in range(c.NPB):
                kk = KS * j
                cst = 64 * (kk >> 3) + 16 * (kk & 3) + 8 * ((kk >> 2) & 1)
                assert (kk & 3) == 0 or KS >= 8, "k0 and KS * j must not share bits"
                e("v_add_u32", self.WB[j][2], cst, t[5])
                e("v_add_u32", self.WB[j][0], c.STAGE, self.WB[j][2])
                e("v_add_u32", self.WB[j][1], 2 * c.STAGE, self.WB[j][2])
            e("v_mul_lo_u32", t[7], k0, st[5])
            e("v_lshl_add_u32", self.vVB[0], px, 4, t[7])
            e("s_mul_i32", st[4], st[5], KS)
            for j in range(1, c.NPB):
                e("v_add_u32", self.vVB[j], st[4], self.vVB[j - 1])
            e("s_mul_i32", self.s_bstep, st[5], c.BK, comment="B advances BK rows per K-tile")

    def kcontig_goff64(self, Voff, NP, ld_bytes):
        """global offsets of a k-contiguous operand's pieces: (xr + 32 i) * ld * 8 + pc * 16, pc = tid % 8, xr = tid / 8 (per run: the K
        tail of a run overwrites them with the out-of-bounds offset)"""
        e, t, st = self.p.emit, self.vt, self.s_t
        e("v_and_b32", t[0], 7, v(0))
        e("v_lshrrev_b32", t[1], 3, v(0))
        e("v_mul_lo_u32", t[7], t[1], ld_bytes)
        e("v_lshl_add_u32", Voff[0], t[0], 4, t[7])
        e("s_lshl_b32", st[4], ld_bytes, 5)                  # 32 rows
        for i in range(1, NP):
            e("v_add_u32", Voff[i], st[4], Voff[i - 1])

    def ab_descriptors(self, again=False):
        """srdA / srdB of the run k in [kb, kb + Keff) of tile (m0, n0).  Clobbers s_t[0], [2], [3], [5]."""
        c, p = self.c, self.p
        e = p.emit
        st = self.s_t
        Keff = self.s_Keff
        e("s_lshl_b32", st[3], self.s_lda, 3, comment="lda * 8 bytes")
        e("s_lshl_b32", st[5], self.s_ldb, 3, comment="ldb * 8 bytes")
        A_, B_ = self.ka0.sub(0, 2), self.ka0.sub(2, 2)
        # A panel: base = A + m0 * lda * 8 + kb * 8; bytes = (min(M - m0, BM) - 1) * lda * 8 + Keff * 8
        e("s_mul_hi_u32", st[2], self.s_m0, st[3])
        e("s_mul_i32", st[0], self.s_m0, st[3])
        e("s_add_u32", self.srdA[0], A_[0], st[0])
        e("s_addc_u32", self.srdA[1], A_[1], st[2])
        if c.persistent:
            e("s_lshl_b32", st[0], self.s_kb, 3)
            e("s_add_u32", self.srdA[0], self.srdA[0], st[0])
            e("s_addc_u32", self.srdA[1], self.srdA[1], 0)
        e("s_and_b32", self.srdA[1], self.srdA[1], 0xffff)
        e("s_sub_u32", st[0], self.s_M, self.s_m0)
        e("s_min_u32", st[0], st[0], c.BM)
        e("s_sub_u32", st[0], st[0], 1)
        e("s_mul_i32", st[0], st[0], st[3])
        e("s_lshl_b32", st[2], Keff, 3)
        e("s_add_u32", self.srdA[2], st[0], st[2])
        e("s_mov_b32", self.srdA[3], 0x00020000)
        if c.b_kcontig:
            # B^T panel: base = B + n0 * ldb * 8 + kb * 8; bytes = (min(N - n0, BN) - 1) * ldb * 8 + Keff * 8
            e("s_mul_hi_u32", st[2], self.s_n0, st[5])
            e("s_mul_i32", st[0], self.s_n0, st[5])
            e("s_add_u32", self.srdB[0], B_[0], st[0])
            e("s_addc_u32", self.srdB[1], B_[1], st[2])
            if c.persistent:
                e("s_lshl_b32", st[0], self.s_kb, 3)
                e("s_add_u32", self.srdB[0], self.srdB[0], st[0])
                e("s_addc_u32", self.srdB[1], self.srdB[1], 0)
            e("s_and_b32", self.srdB[1], self.srdB[1], 0xffff)
            e("s_sub_u32", st[0], self.s_N, self.s_n0)
            e("s_min_u32", st[0], st[0], c.BN)
            e("s_sub_u32", st[0], st[0], 1)
            e("s_mul_i32", st[0], st[0], st[5])
            e("s_lshl_b32", st[2], Keff, 3)
            e("s_add_u32", self.srdB[2], st[0], st[2])
        else:
            # B panel: base = B + n0 * 8 + kb * ldb * 8; bytes = (Keff - 1) * ldb * 8 + (N - n0) * 8
            e("s_lshl_b32", st[0], self.s_n0, 3)
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
            e("s_lshl_b32", st[2], st[2], 3)
            e("s_add_u32", self.srdB[2], st[0], st[2])
        e("s_mov_b32", self.srdB[3], 0x00020000)

    def run_setup(self):
        """one run = k in [kb, kb + Keff) of tile (m0, n0)"""
        c, p = self.c, self.p
        e = p.emit
        t, st = self.vt, self.s_t
        RS = c.RS
        Keff = self.s_Keff
        # K tail (Keff a multiple of 2): A pieces of the last K-tile beyond K read as 0
        e("s_and_b32", self.s_ktail, Keff, c.BK - 1)
        e("v_and_b32", t[5], 7, v(0))
        e("v_lshlrev_b32", t[5], 1, t[5])
        e("v_cmp_gt_u32", self.s_tm, self.s_ktail, t[5])
        e("s_lshl_b32", st[3], self.s_lda, 3, comment="lda * 8 bytes")
        self.kcontig_goff64(self.vVA, c.NPA, st[3])
        e("s_lshl_b32", st[5], self.s_ldb, 3, comment="ldb * 8 bytes")
        if c.b_kcontig:
            self.kcontig_goff64(self.vVB, c.NPB, st[5])
        e("s_nop", 4)
        self.ab_descriptors()
        self.c_descriptor()
        e("s_add_u32", self.s_rem, Keff, c.BK - 1)
        e("s_lshr_b32", self.s_rem, self.s_rem, (c.BK).bit_length() - 1)
        # ---- tile 0 -> LDS stage 0, tile 1 -> staging registers ----
        L_slow, L_join = p.label("fewtiles"), p.label("tiles01")
        fast = not c.debug
        if fast:
            # three or more K-tiles: tiles 0 and 1 requested back to back (tile 0 through the idle fragment registers): a run starts
            # after one memory latency instead of two (f32_kernel.py run_setup)
            state = (list(self.vmq), list(self.lgq))
            e("s_cmp_lt_u32", self.s_rem, 3)
            e("s_cbranch_scc1", L_slow)
            pool = [r for slot in range(2) for r in (self.fa[slot] + self.fb[slot])]
            assert len(pool) >= c.NPA + c.NPB
            real = (self.stA, self.stB)
            tmp = (pool[:c.NPA], pool[c.NPA:c.NPA + c.NPB])
            self.stA, self.stB = tmp
            self.issue_loads_all()
            self.advance_srds()
            self.stA, self.stB = real
            self.issue_loads_all()
            self.advance_srds()
            self.stA, self.stB = tmp
            for pi in range(c.NPA):
                self.store_A_piece(pi, k=2)
            for pj in range(c.NPB):
                self.store_B_piece(pj, k=2)
            self.stA, self.stB = real
            e("s_branch", L_join)
            fast_state = (list(self.vmq), list(self.lgq))
            self.vmq, self.lgq = state
            p.place(L_slow)
        self.tail_mask_if(self.s_rem, 1)
        self.issue_loads_all()
        self.advance_srds()
        if c.debug:
            e("s_waitcnt", vmcnt=0)
            self.vmq.clear()
            for k_ in range(4):
                self.dump(f"stA0[{k_}]", self.stA[0][k_])
            for k_ in range(4):
                self.dump(f"stB0[{k_}]", self.stB[0][k_])
        for pi in range(c.NPA):
            self.store_A_piece(pi, k=2)
        for pj in range(c.NPB):
            self.store_B_piece(pj, k=2)
        if c.debug:
            self.lg_wait(None)
            e("s_barrier")
            for k_ in range(0, 4):
                self.dump_lds(f"lds[{k_ * 1024}+4tid]", k_ * 1024)
            self.dump_lds("ldsB[0+4tid]", c.BM * RS)
            e("s_barrier")
        self.tail_mask_if(self.s_rem, 2)
        self.issue_loads_all()
        self.advance_srds()
        if fast:
            assert (self.vmq, self.lgq) == fast_state, "the two prologue paths must leave the same loads and stores in flight"
            p.place(L_join)
        self.tail_mask_if(self.s_rem, 3)
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
                self.lg_issue(("R", 0))

    def c_descriptor(self):
        """srdC = the whole matrix: bytes = (M - 1) * ldc * 8 + N * 8"""
        e, st = self.p.emit, self.s_t
        C_ = self.ka0.sub(4, 2)
        e("s_lshl_b32", self.s_ldc4, self.s_ldc, 3)
        e("s_mov_b32", self.srdC[0], C_[0])
        e("s_and_b32", self.srdC[1], C_[1], 0xffff)
        e("s_sub_u32", st[0], self.s_M, 1)
        e("s_mul_i32", st[0], st[0], self.s_ldc4)
        e("s_lshl_b32", st[2], self.s_N, 3)
        e("s_add_u32", self.srdC[2], st[0], st[2])
        e("s_mov_b32", self.srdC[3], 0x00020000)
        e("s_lshl_b32", self.s_ldc20, self.s_ldc4, 2)

    def issue_loads_all(self):
        for pi in range(self.c.NPA):
            self.load_A_piece(pi)
        for pj in range(self.c.NPB):
            self.load_B_piece(pj)

    def advance_srds(self, which=None):
        e = self.p.emit
        ops = []
        for srd, step in ((self.srdA, self.c.BK * 8), (self.srdB, self.c.BK * 8 if self.c.b_kcontig else self.s_bstep)):
            ops += [("s_add_u32", srd[0], srd[0], step), ("s_addc_u32", srd[1], srd[1], 0),
                    ("s_sub_u32", srd[2], srd[2], step), ("s_cselect_b32", srd[2], 0, srd[2])]
        if which is None:
            for o in ops:
                e(*o)
        return ops

    def mask_last_pieces_if(self, sreg, value):
        pass      # (K is even: the 16-byte pieces of 2 doubles are all-or-nothing)

    def apply_tail_mask(self):
        for r in list(self.vVA) + (list(self.vVB) if self.c.b_kcontig else []):
            self.p.emit("v_cndmask_b32", r, self.v_oob, r, self.s_tm)

    # ------------------------------------------------------------------ LDS stores: one ds_write2_b64 per piece
    def store_A_piece(self, pi, ops=None, k=0):
        r = self.stA[pi]
        out = [("vmwait", ("A", pi)),
               ("ldsw", "ds_write2_b64", (self.WA[pi][k], r.sub(0, 2), r.sub(2, 2)), {"offset0": 0, "offset1": 2})]
        if ops is None:
            self.run_ops(out)
        return out

    def store_B_piece(self, pj, ops=None, k=0):
        r = self.stB[pj]
        out = [("vmwait", ("B", pj)),
               ("ldsw", "ds_write2_b64", (self.WB[pj][k], r.sub(0, 2), r.sub(2, 2)),
                {"offset0": 0, "offset1": 2 if self.c.b_kcontig else self.c.RS // 8})]
        if ops is None:
            self.run_ops(out)
        return out

    def staging_ops(self, wr_k):
        c, stg = self.c, []
        for pi in range(c.NPA):
            stg += self.store_A_piece(pi, ops=[], k=wr_k)
            stg.append(("loadA", pi))
        for pj in range(c.NPB):
            stg += self.store_B_piece(pj, ops=[], k=wr_k)
            stg.append(("loadB", pj))
        return stg

    # ------------------------------------------------------------------ matrix instruction, slice fold
    def emit_mfma(self, b, slot, i, n, u, srcc):
        self.p.emit("v_mfma_f64_16x16x4_f64", self.acc[b], self.fa[slot][i].sub(2 * u, 2), self.fb[slot][n].sub(2 * u, 2), srcc)

    def fold_before(self, b):
        T = self.vT[0]
        for r in range(8):
            self.p.emit("v_accvgpr_read_b32", T[r], self.acc[b][r])

    def fold_after(self, b):
        p, e, T = self.p, self.p.emit, self.vT[0]
        # run += alpha * slice, unfused; alpha == 1: the multiplies (out of line) are a branch not taken
        lmul, lback = p.label("amul"), p.label("aback")
        e("s_cmp_lg_u32", self.s_a1, 0)
        e("s_cbranch_scc1", lmul)
        p.place(lback)
        self.outlined.append((lmul, [("v_mul_f64", T.sub(2 * d, 2), self.s_al, T.sub(2 * d, 2)) for d in range(4)], lback))
        if self.c.runv:
            for d in range(4):
                e("v_add_f64", self.run[b].sub(2 * d, 2), self.run[b].sub(2 * d, 2), T.sub(2 * d, 2))
            return
        for d in range(4):
            tt = T.sub(8 + 2 * (d % 2), 2)
            e("v_accvgpr_read_b32", tt[0], self.run[b][2 * d])
            e("v_accvgpr_read_b32", tt[1], self.run[b][2 * d + 1])
            e("v_add_f64", tt, tt, T.sub(2 * d, 2))
            e("v_accvgpr_write_b32", self.run[b][2 * d], tt[0])
            e("v_accvgpr_write_b32", self.run[b][2 * d + 1], tt[1])

    def init_accumulators(self):
        if not self.c.runv:
            return Gen.init_accumulators(self)
        from .f32_kernel import MODE_NORMAL
        c, p = self.c, self.p
        e = p.emit
        for b in range(c.NB):
            for r in range(c.ACCR):
                e("v_accvgpr_write_b32", self.acc[b][r], 0)
        keep = p.label("keeprun")
        if c.persistent:
            e("s_cmp_lg_u32", self.s_mode, MODE_NORMAL)
            e("s_cbranch_scc1", keep)
        for b in range(c.NB):
            for d in range(4):
                e("v_mov_b64", self.run[b].sub(2 * d, 2), 0)
        self.load_beta_c()
        p.place(keep)

    def fold_block(self, b):
        e, T = self.p.emit, self.vT[0]
        for r in range(8):
            e("v_accvgpr_read_b32", T[r], self.acc[b][r])
        for d in range(4):
            e("v_mul_f64", T.sub(2 * d, 2), self.s_al, T.sub(2 * d, 2))      # (1.0 * x is x)
        if self.c.runv:
            for d in range(4):
                e("v_add_f64", self.run[b].sub(2 * d, 2), self.run[b].sub(2 * d, 2), T.sub(2 * d, 2))
            return
        for d in range(4):
            tt = T.sub(8 + 2 * (d % 2), 2)
            e("v_accvgpr_read_b32", tt[0], self.run[b][2 * d])
            e("v_accvgpr_read_b32", tt[1], self.run[b][2 * d + 1])
            e("v_add_f64", tt, tt, T.sub(2 * d, 2))
            e("v_accvgpr_write_b32", self.run[b][2 * d], tt[0])
            e("v_accvgpr_write_b32", self.run[b][2 * d + 1], tt[1])

    def acc_add_block(self, b):
        e, T = self.p.emit, self.vT[0]
        for d in range(4):
            tt = T.sub(8 + 2 * (d % 2), 2)
            e("v_accvgpr_read_b32", tt[0], self.acc[b][2 * d])
            e("v_accvgpr_read_b32", tt[1], self.acc[b][2 * d + 1])
            e("v_add_f64", tt, tt, T.sub(2 * d, 2))
            e("v_accvgpr_write_b32", self.acc[b][2 * d], tt[0])
            e("v_accvgpr_write_b32", self.acc[b][2 * d + 1], tt[1])

    # ------------------------------------------------------------------ pipelined tile transitions (f32_kernel.py Cfg.pipe)
    def trans_after(self, b):
        """transition body, block b = (i, n): vT[0..7] hold the slice sum (4 doubles: rows q + 4 d of the block) of the tile being
        FINISHED, the MFMA in front of this gap has restarted the chain for the new tile: C = run + alpha * slice (one chain: alpha *
        sum) leaves for memory from here, the running sum is zeroed for the new tile (beta == 0 in pipelined launches).  Rows (+ 4 d,
        + 16 i) through the stores' scalar offset from srdCd = this wave's first row of the old tile."""
        c, p, e, T, st = self.c, self.p, self.p.emit, self.vT[0], self.s_t
        i, n = b // c.TN, b % c.TN
        assert c.runv or not c.exact
        lmul, lback = p.label("tmul"), p.label("tback")
        e("s_cmp_lg_u32", self.s_a1, 0)
        e("s_cbranch_scc1", lmul)
        p.place(lback)
        self.outlined.append((lmul, [("v_mul_f64", T.sub(2 * d, 2), self.s_al, T.sub(2 * d, 2)) for d in range(4)], lback))
        soff = st[0]
        for d in range(4):
            if c.exact:
                tt = T.sub(8 + 2 * (d % 2), 2)
                e("v_add_f64", tt, self.run[b].sub(2 * d, 2), T.sub(2 * d, 2))
                e("v_mov_b64", self.run[b].sub(2 * d, 2), 0)
            else:
                tt = T.sub(2 * d, 2)
            if 16 * i + 4 * d:
                e("s_mul_i32", soff, self.s_ldc4, 16 * i + 4 * d)
            else:
                e("s_mov_b32", soff, 0)
            e("buffer_store_dwordx2", tt, self.vC[n], self.srdCd, soff, offen=True)
            self.vm_issue(("S", b, d))

    def pipe_c_addr(self):
        """(vC, srdCd) for the deferred stores of the tile (m0, n0): srdCd starts at this wave's first row, vC[n] = the lane's offset
        from there (q rows down, block column n; out of bounds beyond N)"""
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        lane, r16, q = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, v(0))
        e("v_and_b32", r16, 15, lane)
        e("v_lshrrev_b32", q, 4, lane)
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
        e("v_mul_lo_u32", t[3], q, self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], r16)
        e("v_lshl_add_u32", t[3], t[4], 3, t[3])
        for n in range(c.TN):
            e("v_add_u32", t[5], 16 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            e("v_add_u32", t[6], 128 * n, t[3])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[6], VCC)

    # ------------------------------------------------------------------ epilogue
    def c_addr_setup(self):
        """vC[n] = byte offset in C of D[q][r16] of block column n: row m0 + wm0 + q (+ 4 per accumulator element, + 16 per
        block row), col n0 + wn0 + r16 + 16n"""
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        lane, r16, q = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, v(0))
        e("v_and_b32", r16, 15, lane)
        e("v_lshrrev_b32", q, 4, lane)
        e("s_add_u32", st[0], self.s_m0, self.s_wm0)
        e("v_add_u32", t[3], st[0], q)
        e("v_mul_lo_u32", t[3], t[3], self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], r16)
        e("v_lshl_add_u32", t[3], t[4], 3, t[3])
        for n in range(c.TN):
            e("v_add_u32", t[5], 16 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            e("v_add_u32", t[6], 128 * n, t[3])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[6], VCC)

    def c_walk(self, fn):
        """visit this lane's accumulator elements in C order: fn(i, d, n) with vC[n] addressing D[q + 4d][r16] of block (i, n)"""
        c = self.c
        for i in range(c.TM):
            for d in range(4):
                fn(i, d)
                if not (i == c.TM - 1 and d == 3):
                    for n in range(c.TN):
                        self.p.emit("v_add_u32", self.vC[n], self.s_ldc20, self.vC[n])

    def load_beta_c(self):
        """laser-order kernels, beta != 0: the running sum starts as beta * C0 (one rounding), loaded through the idle fragment
        registers and drained here, so the loop's counted waits (computed for beta == 0) stay correct"""
        c, p = self.c, self.p
        e = p.emit
        skip = p.label("nobeta")
        e("s_cmp_eq_u32", self.s_b0, 0)
        e("s_cbranch_scc1", skip)
        self.c_addr_setup()
        pool = [r.sub(2 * h, 2) for slot in range(2) for r in (self.fa[slot] + self.fb[slot]) for h in range(2)]
        assert len(pool) >= c.TN

        def row(i, d):
            for n in range(c.TN):
                # (runv: straight into the running sum's own registers)
                dst = self.run[i * c.TN + n].sub(2 * d, 2) if c.runv else pool[n]
                e("buffer_load_dwordx2", dst, self.vC[n], self.srdC, 0, offen=True)
            e("s_waitcnt", vmcnt=0)
            for n in range(c.TN):
                if c.runv:
                    x = self.run[i * c.TN + n].sub(2 * d, 2)
                    e("v_mul_f64", x, self.s_be, x)
                    continue
                e("v_mul_f64", pool[n], self.s_be, pool[n])
                e("v_accvgpr_write_b32", self.run[i * c.TN + n][2 * d], pool[n][0])
                e("v_accvgpr_write_b32", self.run[i * c.TN + n][2 * d + 1], pool[n][1])
        self.c_walk(row)
        p.place(skip)

    def epilogue(self):
        """C = beta * C0 + alpha * (slice sums in order) -- gemm_ukernel_generic.nim:53-76 -- predicated by the descriptor's bounds check"""
        c, p = self.c, self.p
        e = p.emit
        e("s_nop", 15)
        e("s_nop", 7)
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        self.vmq.clear()
        self.lgq.clear()
        if c.debug:
            for k_ in range(4):
                self.dump(f"acc[0][{k_}]", self.acc[0][k_])
        self.mode_dispatch()
        self.c_addr_setup()
        T = self.vT[0]

        def read_acc(tt, b, d):
            e("v_accvgpr_read_b32", tt[0], self.acc[b][2 * d])
            e("v_accvgpr_read_b32", tt[1], self.acc[b][2 * d + 1])
            e("v_mul_f64", tt, self.s_al, tt)                     # (1.0 * x is x)

        if c.exact:
            def row(i, d):
                for n in range(c.TN):
                    b = i * c.TN + n
                    tt, uu = T.sub(4 * (n % 2), 2), T.sub(4 * (n % 2) + 2, 2)
                    read_acc(tt, b, d)
                    if c.runv:
                        e("v_add_f64", tt, self.run[b].sub(2 * d, 2), tt)
                    else:
                        e("v_accvgpr_read_b32", uu[0], self.run[b][2 * d])
                        e("v_accvgpr_read_b32", uu[1], self.run[b][2 * d + 1])
                        e("v_add_f64", tt, uu, tt)
                    e("buffer_store_dwordx2", tt, self.vC[n], self.srdC, 0, offen=True)
            self.c_walk(row)
        else:
            withc, done = p.label("beta"), p.label("stored")
            e("s_cmp_lg_u32", self.s_b0, 0)
            e("s_cbranch_scc1", withc)

            def row0(i, d):
                for n in range(c.TN):
                    tt = T.sub(4 * (n % 2), 2)
                    read_acc(tt, i * c.TN + n, d)
                    e("buffer_store_dwordx2", tt, self.vC[n], self.srdC, 0, offen=True)
            self.c_walk(row0)
            e("s_branch", done)
            p.place(withc)
            self.c_addr_setup()

            def row1(i, d):
                for n in range(c.TN):
                    e("buffer_load_dwordx2", T.sub(8 + 2 * n, 2), self.vC[n], self.srdC, 0, offen=True)
                e("s_waitcnt", vmcnt=0)
                for n in range(c.TN):
                    tt, x = T.sub(2 * n, 2), T.sub(8 + 2 * n, 2)
                    e("v_mul_f64", x, self.s_be, x)
                    read_acc(tt, i * c.TN + n, d)
                    e("v_add_f64", tt, x, tt)
                    e("buffer_store_dwordx2", tt, self.vC[n], self.srdC, 0, offen=True)
            self.c_walk(row1)
            p.place(done)
        self.end_run()


def make(name, **over):
    kw = dict(CONFIGS[name])
    kw.update(over)
    return Gen64(Cfg(name, dtype="f64", **kw))


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
        sym = "lh_f64_" + name
        with open(os.path.join(args.out, sym + ".s"), "w") as f:
            f.write(kernel_text(g, sym))
        print(sym, len(g.p.ins), "instructions")
