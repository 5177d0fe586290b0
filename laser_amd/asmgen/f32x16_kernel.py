"""float32 GEMM kernels on 16x16 blocks (gfx950 assembly): the tile family whose sides are multiples of 32 instead of 64 -- 96x96 and
160x96 -- for the problems the 32x32-block tiles quantise badly.  The reference's own float32 benchmark shape is one of them
(benchmarks/gemm/gemm_bench_float32.nim:383-410, M = N = K = 1920: 225 tiles of 128x128 leave 31 of 256 CUs idle, 240 tiles of
160x96 leave 16), 1536^3 is another (144 tiles of 128x128; 256 tiles of 96x96 = exactly one per CU).  Laser picks its micro-tile per
CPU the same way (gemm_tiling.nim:147-219: MR x NR from the register file); here the tile is picked per problem.

  v_mfma_f32_16x16x4_f32   16x16 block, 4 k per instruction (32 cycles per SIMD, 40 dependent), lane l feeds A[l % 16][l / 16] and
                           B[l / 16][l % 16] and holds D[4 * (l / 16) + d][l % 16], d = 0..3: 4 accumulator registers per block.  An
                           fmaf chain over its 4 k, ascending, bitwise (MI355X_MICROARCH.md) -- the chain of gemm_ukernel_generic.nim:56-66
  kc = 512                 gemm_tiling.nim:310: the running sum is folded every 16 K-tiles of 32 (gemm.nim:150-158)

Program structure: f32_kernel.Gen's (3-stage LDS ring, one barrier per K-tile, one straight-line body per stage with every memory
operation assigned to a gap between two MFMAs and counted waits, accumulators + running sum in AGPRs, persistent scheduler with
K-slice hand-overs), with the f64 kernels' padded-row LDS image (f64_kernel.py) at 4-byte elements:
  a row holds BK = 32 floats as 8 chunks of 16 bytes, chunk 4 * (k / 16) + k % 4 holding k, k + 4, k + 8, k + 12 -- the four k-steps
  lane group q = k % 4 needs from ONE ds_read_b128 -- and is 144 bytes long (9 chunks: the 16 rows a 16-lane group reads fall into
  16 different 4-bank groups, no swizzle).  Stores are two ds_write2_b32 per 16-byte piece:
  A piece (k0 .. k0 + 3 of one row, k0 = 4 pc)   -> the same word of chunks 4g .. 4g + 3:   +0, +16, +32, +48 bytes
  B piece (columns x0 .. x0 + 3 of one k)        -> the same word of rows x0 .. x0 + 3:      +0, +144, +288, +432 bytes
Operands: A row-major (k-contiguous), B row-major or passed transposed (`_nt`), C with any positive column stride (rows slow); K a multiple
of 4 (16-byte pieces are all-or-nothing); any alpha / beta; fused epilogue C = act(.. + bias) with a strided bias view and relu like the
32x32-block kernels (README.md:238-242 of the reference plans this fusion); no fused prologue."""
from .core import v, s, VCC
from .f32_kernel import Gen, Cfg, kernel_text, KA_A, KA_LDA, KA_CONV1, KA_BSA, KA_BIAS, KA_EPI  # noqa: F401
from .f64_kernel import Gen64

CONFIGS = {
    # 2 x 2 waves of 48x48 = 9 blocks: 36 accumulator registers (+ 36 for the running sum); 81 KiB of LDS
    "exact_96x96x32": dict(BM=96, BN=96, BK=32, exact=True, runv=True),
    "fast_96x96x32": dict(BM=96, BN=96, BK=32, exact=False),
    "exact_96x96x32_nt": dict(BM=96, BN=96, BK=32, exact=True, b_kcontig=True, runv=True),
    "fast_96x96x32_nt": dict(BM=96, BN=96, BK=32, exact=False, b_kcontig=True),
    # 2 x 2 waves of 80x48 = 15 blocks: 60 + 60 registers; 108 KiB of LDS
    "exact_160x96x32": dict(BM=160, BN=96, BK=32, exact=True, runv=True),
    "fast_160x96x32": dict(BM=160, BN=96, BK=32, exact=False),
    "exact_160x96x32_nt": dict(BM=160, BN=96, BK=32, exact=True, b_kcontig=True, runv=True),
    "fast_160x96x32_nt": dict(BM=160, BN=96, BK=32, exact=False, b_kcontig=True),
    # more sides for more problems (a tile is picked per problem, gemm_f32_asm.cpp): 128x96 (2 x 2 waves of 64x48 = 12 blocks:
    # 1000x3000x2000 is 256 tiles of it), 192x96 (96x48 = 18 blocks: 3072^3 is 512 tiles = two exact rounds), 160x160 (80x80 = 25
    # blocks, 100 + 100 accumulator registers, 135 KiB of LDS: 2560^3 is 256 tiles, 5120^3 four exact rounds)
    # (laser-order: the running sum in arch VGPRs; the two largest tiles move fragments + staging to AGPRs to make room)
    **{f"{ex}_{bm}x{bn}x32{nt}": dict(BM=bm, BN=bn, BK=32, exact=(ex == "exact"), **({"b_kcontig": True} if nt else {}),
                                      **({"runv": True, "dataa": bm >= 160} if ex == "exact" else {}))
       for bm, bn in ((128, 96), (192, 96), (160, 160)) for nt in ("", "_nt") for ex in ("exact", "fast")},
}


class Gen16(Gen64):
    ab_descriptors = Gen.ab_descriptors      # 4-byte elements: the f32 kernels' descriptors (run_setup and the pipelined switch)

    # ------------------------------------------------------------------ registers
    def alloc(self):
        c, p = self.c, self.p
        S, V = p.salloc, p.valloc
        assert c.BK == 32 and c.BM % 32 == 0 and c.BN % 32 == 0 and not c.debug
        self.ka0 = S(8, align=4)
        self.ka1 = S(8, align=4)
        self.s_lda, self.s_ldb, self.s_ldc, self.s_M, self.s_N, self.s_K = (self.ka1[i] for i in range(6))
        self.s_alpha, self.s_beta = self.ka1[6], self.ka1[7]     # float32 bit patterns
        self.srdA, self.srdB, self.srdC = S(4), S(4), S(4)
        self.s_rem, self.s_cnt = S(), S()
        self.s_bstep = S()
        self.s_m0, self.s_n0, self.s_wave, self.s_wm0, self.s_wn0 = S(), S(), S(), S(), S()
        self.s_t = [S() for _ in range(6)]
        self.s_bsA, self.s_bsBC = S(2, align=2), S(4, align=4)   # batch strides in bytes (grid y = batch index)
        self.s_ldc4 = S()
        self.s_csC4 = S()                      # column stride of C in bytes (KA_EPI + 12; 0 in the arguments = dense)
        self.alloc_sched()
        self.srdBias = S(4)                    # fused epilogue: the bias view (base, -, bytes, flags)
        self.s_epi = S(4, align=4)             # rowStrideBias, colStrideBias (elements), activation, -
        if c.pipe:
            # pipelined tile transitions (f32_kernel.py Cfg.pipe): bit 0 = this launch may pipeline, bit 1 = armed
            self.s_pipe = S()
            self.srdCd = S(4)        # C from this wave's first row of the tile being finished
        self.acc = [p.aalloc(4) for _ in range(c.NB)]
        # runv (f32_kernel.py Cfg): the running sum in arch VGPRs -- the slice fold is 4 v_accvgpr_read + 2 v_pk_add_f32 per block where
        # the all-AGPR plan has 16 VALU operations; dataa: fragments + staging in AGPRs (the tiles whose VGPR file would overflow)
        self.run = [V(4) if c.runv else p.aalloc(4) for _ in range(c.NB)] if c.exact else None
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
        self.ndump = 0
        self.dump_names = []
        self.vT = [V(16)]
        blk = V(12, align=4)
        self.vt = [blk[i] for i in range(10)]
        self.vF, self.vFaddr, self.vFoff = blk.sub(4, 4), blk[10], blk[11]

    # ------------------------------------------------------------------ prologue (f32_kernel.Gen.prologue: once, scheduler, run_setup)
    def once(self):
        c, p = self.c, self.p
        e = p.emit
        t, st = self.vt, self.s_t
        RS = c.RS
        p.note(f"f32 (16x16 blocks) {c.name}: {c.BM}x{c.BN}x{c.BK} tile, 4 waves, wave tile {c.WTM}x{c.WTN}, "
               f"{'laser-order (kc = 512 slices)' if c.exact else 'one accumulation chain'}")
        e("s_load_dwordx8", self.ka0, s(0, 2), KA_A)
        e("s_load_dwordx8", self.ka1, s(0, 2), KA_LDA)
        # batched problems: workgroup id y = batch index, operand b at base + b * batch stride (bytes; 0 for plain launches) -- the
        # f32 kernels' argument slots
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
        if c.pipe:
            # tile transitions of this launch may be pipelined: beta == 0 (the next tile's running sum starts at 0), K a multiple of BK
            # (no K-tail masks to undo between tiles), at least three K-tiles (the switch happens two tile bodies before a tile's end)
            e("s_load_dwordx2", self.srdBias.sub(0, 2), s(0, 2), KA_BIAS)
            e("s_load_dword", st[2], s(0, 2), KA_EPI + 8)
            e("s_waitcnt", lgkmcnt=0)
            e("s_or_b32", st[3], self.srdBias[0], self.srdBias[1])
            e("s_or_b32", st[3], st[3], st[2])             # (a bias pointer or an activation: the fused epilogue, not a transition body)
            e("s_and_b32", st[4], self.s_beta, 0x7fffffff)
            e("s_or_b32", st[3], st[3], st[4])
            e("s_and_b32", st[4], self.s_K, c.BK - 1)
            e("s_or_b32", st[3], st[3], st[4])
            e("s_cmp_eq_u32", st[3], 0)
            e("s_cselect_b32", self.s_pipe, 1, 0)
            e("s_cmp_lt_u32", self.s_K, 3 * c.BK)
            e("s_cselect_b32", self.s_pipe, 0, self.s_pipe)
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
        # fragment reads of group g (16 k): (wm0 + r16) * RS [+ BM * RS + (wn0 + r16) * RS for B] + (4g + q) * 16
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
        # k-contiguous pieces (A always; B when it is passed transposed): piece column pc = tid % 8 (k0 = 4 pc), row xr = tid / 8
        # (+ 32 per piece);  LDS: row * RS + (pc >> 2) * 64 + (pc & 3) * 4, element j of the piece 16 j bytes further
        e("v_lshrrev_b32", t[5], 2, pc)
        e("v_lshlrev_b32", t[5], 6, t[5])                    # (pc >> 2) * 64
        e("v_and_b32", t[6], 3, pc)
        e("v_lshl_add_u32", t[5], t[6], 2, t[5])             # + (pc & 3) * 4
        e("v_mul_u32_u24", t[6], RS, xr)
        e("v_add_u32", t[5], t[5], t[6])
        for i in range(c.NPA):
            e("v_add_u32", self.WA[i][2], 32 * RS * i, t[5])
            e("v_add_u32", self.WA[i][0], c.STAGE, self.WA[i][2])
            e("v_add_u32", self.WA[i][1], 2 * c.STAGE, self.WA[i][2])
        e("s_lshl_b32", st[5], self.s_ldb, 2, comment="ldb * 4 bytes")
        if c.b_kcontig:
            e("v_add_u32", t[5], c.BM * RS, t[5])
            for j in range(c.NPB):
                e("v_add_u32", self.WB[j][2], 32 * RS * j, t[5])
                e("v_add_u32", self.WB[j][0], c.STAGE, self.WB[j][2])
                e("v_add_u32", self.WB[j][1], 2 * c.STAGE, self.WB[j][2])
            e("s_mov_b32", self.s_bstep, c.BK * 4)
        else:
            # B pieces (x-contiguous: columns x0 .. x0 + 3 of one k): column quad c8 = tid % 8, row k = tid / 8 (all 32 k of the
            # K-tile in one pass), pass j covers columns 32 j .. 32 j + 31 (8 lanes read 128 contiguous bytes of a row of B)
            #   LDS: (BM + 32 j + 4 c8) * RS + (k >> 4) * 64 + (k & 3) * 16 + ((k >> 2) & 3) * 4, element e of the piece e * RS further
            c8, k = t[0], t[1]
            e("v_lshrrev_b32", t[5], 4, k)
            e("v_lshlrev_b32", t[5], 6, t[5])                    # (k >> 4) * 64
            e("v_and_b32", t[6], 3, k)
            e("v_lshl_add_u32", t[5], t[6], 4, t[5])             # + (k & 3) * 16
            e("v_bfe_u32", t[6], k, 2, 2)
            e("v_lshl_add_u32", t[5], t[6], 2, t[5])             # + ((k >> 2) & 3) * 4
            e("v_mul_u32_u24", t[6], 4 * RS, c8)
            e("v_add_u32", t[5], t[5], t[6])
            e("v_add_u32", t[5], c.BM * RS, t[5])
            for j in range(c.NPB):
                e("v_add_u32", self.WB[j][2], 32 * RS * j, t[5])
                e("v_add_u32", self.WB[j][0], c.STAGE, self.WB[j][2])
                e("v_add_u32", self.WB[j][1], 2 * c.STAGE, self.WB[j][2])
            e("v_mul_lo_u32", t[7], k, st[5])
            e("v_lshl_add_u32", self.vVB[0], c8, 4, t[7])
            for j in range(1, c.NPB):
                e("v_add_u32", self.vVB[j], 128, self.vVB[j - 1])
            e("s_mul_i32", self.s_bstep, st[5], c.BK, comment="B advances BK rows per K-tile")

    def run_setup(self):
        """one run = k in [kb, kb + Keff) of tile (m0, n0) (Keff a multiple of 4)"""
        c, p = self.c, self.p
        e = p.emit
        t, st = self.vt, self.s_t
        Keff = self.s_Keff
        # K tail: pieces of the last K-tile beyond K get an offset the bounds check rejects (they read as 0, like Laser's zero-padded
        # panels, gemm_packing.nim:46-55)
        e("s_and_b32", self.s_ktail, Keff, c.BK - 1)
        e("v_and_b32", t[5], 7, v(0))
        e("v_lshlrev_b32", t[5], 2, t[5])
        e("v_cmp_gt_u32", self.s_tm, self.s_ktail, t[5])
        e("s_lshl_b32", st[3], self.s_lda, 2, comment="lda * 4 bytes")
        self.kcontig_goff64(self.vVA, c.NPA, st[3])          # (xr + 32 i) * ld bytes + pc * 16: the same piece map as the f64 kernels
        e("s_lshl_b32", st[5], self.s_ldb, 2, comment="ldb * 4 bytes")
        if c.b_kcontig:
            self.kcontig_goff64(self.vVB, c.NPB, st[5])
        e("s_nop", 4)
        Gen.ab_descriptors(self)                             # 4-byte elements: the f32 kernels' descriptors
        self.c_descriptor()
        e("s_add_u32", self.s_rem, Keff, c.BK - 1)
        e("s_lshr_b32", self.s_rem, self.s_rem, (c.BK).bit_length() - 1)
        # ---- tile 0 -> LDS stage 0, tile 1 -> staging registers ----
        # three or more K-tiles: tiles 0 and 1 requested back to back (tile 0 through the idle fragment registers): a run starts after
        # one memory latency instead of two (f32_kernel.py run_setup)
        L_slow, L_join = p.label("fewtiles"), p.label("tiles01")
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
        for pi in range(c.NPA):
            self.store_A_piece(pi, k=2)
        for pj in range(c.NPB):
            self.store_B_piece(pj, k=2)
        self.tail_mask_if(self.s_rem, 2)
        self.issue_loads_all()
        self.advance_srds()
        assert (self.vmq, self.lgq) == fast_state, "the two prologue paths must leave the same loads and stores in flight"
        p.place(L_join)
        self.tail_mask_if(self.s_rem, 3)
        self.init_accumulators()
        self.lg_wait(None)
        e("s_barrier")
        self.read_group(0, 0, 0)

    def c_descriptor(self):
        """srdC = the whole matrix: bytes = (M - 1) * ldc * 4 + N * 4"""
        e, st = self.p.emit, self.s_t
        C_ = self.ka0.sub(4, 2)
        e("s_lshl_b32", self.s_ldc4, self.s_ldc, 2)
        e("s_mov_b32", self.srdC[0], C_[0])
        e("s_and_b32", self.srdC[1], C_[1], 0xffff)
        e("s_sub_u32", st[0], self.s_M, 1)
        e("s_mul_i32", st[0], st[0], self.s_ldc4)
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

    def advance_srds(self, which=None):
        return Gen.advance_srds(self, which)                 # steps of BK * 4 bytes (A, B^T) / BK rows (B)

    # ------------------------------------------------------------------ LDS stores: two ds_write2_b32 per piece
    def store_A_piece(self, pi, ops=None, k=0):
        r = self.stA[pi]
        out = [("vmwait", ("A", pi)),
               ("ldsw", "ds_write2_b32", (self.WA[pi][k], r[0], r[1]), {"offset0": 0, "offset1": 4}),
               ("ldsw", "ds_write2_b32", (self.WA[pi][k], r[2], r[3]), {"offset0": 8, "offset1": 12})]
        if ops is None:
            self.run_ops(out)
        return out

    def store_B_piece(self, pj, ops=None, k=0):
        r = self.stB[pj]
        u = 4 if self.c.b_kcontig else self.c.RS // 4      # dwords between the piece's elements in LDS
        out = [("vmwait", ("B", pj)),
               ("ldsw", "ds_write2_b32", (self.WB[pj][k], r[0], r[1]), {"offset0": 0, "offset1": u}),
               ("ldsw", "ds_write2_b32", (self.WB[pj][k], r[2], r[3]), {"offset0": 2 * u, "offset1": 3 * u})]
        if ops is None:
            self.run_ops(out)
        return out

    # ------------------------------------------------------------------ matrix instruction, slice fold
    def emit_mfma(self, b, slot, i, n, u, srcc):
        self.p.emit("v_mfma_f32_16x16x4_f32", self.acc[b], self.fa[slot][i][u], self.fb[slot][n][u], srcc)

    def fold_before(self, b):
        T = self.vT[0]
        for r in range(4):
            self.p.emit("v_accvgpr_read_b32", T[r], self.acc[b][r])

    def fold_after(self, b):
        p, e, T = self.p, self.p.emit, self.vT[0]
        # run += alpha * slice, unfused (gemm_ukernel_generic.nim:68-76); alpha == 1: the multiplies (out of line) are a branch not taken
        lmul, lback = p.label("amul"), p.label("aback")
        e("s_cmp_lg_u32", self.s_alpha, 0x3f800000)
        e("s_cbranch_scc1", lmul)
        p.place(lback)
        self.outlined.append((lmul, [("v_mul_f32", T[r], self.s_alpha, T[r]) for r in range(4)], lback))
        if self.c.runv:
            for j in range(2):
                e("v_pk_add_f32", self.run[b].sub(2 * j, 2), self.run[b].sub(2 * j, 2), T.sub(2 * j, 2))
            return
        for r in range(4):
            tt = self.vt[r]
            e("v_accvgpr_read_b32", tt, self.run[b][r])
            e("v_add_f32", tt, tt, T[r])
            e("v_accvgpr_write_b32", self.run[b][r], tt)

    def trans_after(self, b):
        """transition body (Cfg.pipe), block b = (i, n): vT[0..3] hold the slice sum of the tile being FINISHED, the MFMA in front of this
        gap has restarted the chain for the new tile: C = run + alpha * slice (one chain: alpha * sum) leaves for memory from here, the
        running sum is zeroed for the new tile (beta == 0 in pipelined launches).  Rows (+ d, + 16 i) through the stores' scalar offset
        from srdCd = this wave's first row of the old tile."""
        c, p, e, T, st = self.c, self.p, self.p.emit, self.vT[0], self.s_t
        i, n = b // c.TN, b % c.TN
        lmul, lback = p.label("tmul"), p.label("tback")
        e("s_cmp_lg_u32", self.s_alpha, 0x3f800000)
        e("s_cbranch_scc1", lmul)
        p.place(lback)
        self.outlined.append((lmul, [("v_mul_f32", T[r], self.s_alpha, T[r]) for r in range(4)], lback))
        soff = st[0]
        pair = v(self.vt[0].idx, 4)
        for d in range(4):
            if c.exact:
                if c.runv:
                    if d % 2 == 0:
                        e("v_pk_add_f32", pair.sub(d, 2), self.run[b].sub(d, 2), T.sub(d, 2))
                        e("v_mov_b64", self.run[b].sub(d, 2), 0)
                    tt = pair[d]
                else:
                    tt = self.vt[d]
                    e("v_accvgpr_read_b32", tt, self.run[b][d])
                    e("v_add_f32", tt, tt, T[d])
                    e("v_accvgpr_write_b32", self.run[b][d], 0)
            else:
                tt = T[d]
            if 16 * i + d:
                e("s_mul_i32", soff, self.s_ldc4, 16 * i + d)
            else:
                e("s_mov_b32", soff, 0)
            e("buffer_store_dword", tt, self.vC[n], self.srdCd, soff, offen=True)
            self.vm_issue(("S", b, d))

    def pipe_c_addr(self):
        """(vC, srdCd) for the deferred stores of the tile (m0, n0): srdCd starts at this wave's first row, vC[n] = the lane's offset
        from there (4 q rows down, block column n; out of bounds beyond N)"""
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
        e("v_lshlrev_b32", t[3], 2, q)
        e("v_mul_lo_u32", t[3], t[3], self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], r16)
        e("v_mul_lo_u32", t[6], t[4], self.s_csC4)
        e("v_add_u32", t[3], t[3], t[6])
        e("s_lshl_b32", st[1], self.s_csC4, 4)
        for n in range(c.TN):
            e("v_add_u32", t[5], 16 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            if n:
                e("v_add_u32", t[3], st[1], t[3])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[3], VCC)

    def init_accumulators(self):
        if not self.c.runv:
            return Gen.init_accumulators(self)
        # (f32_kernel.Gen.init_accumulators with the running sum in arch VGPRs)
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
            for j in range(2):
                e("v_mov_b64", self.run[b].sub(2 * j, 2), 0)
        self.load_beta_c()
        p.place(keep)

    def fold_block(self, b):
        e, T = self.p.emit, self.vT[0]
        for r in range(4):
            e("v_accvgpr_read_b32", T[r], self.acc[b][r])
        for r in range(4):
            e("v_mul_f32", T[r], self.s_alpha, T[r])      # (1.0 * x is x)
        if self.c.runv:
            for j in range(2):
                e("v_pk_add_f32", self.run[b].sub(2 * j, 2), self.run[b].sub(2 * j, 2), T.sub(2 * j, 2))
            return
        for r in range(4):
            tt = self.vt[r]
            e("v_accvgpr_read_b32", tt, self.run[b][r])
            e("v_add_f32", tt, tt, T[r])
            e("v_accvgpr_write_b32", self.run[b][r], tt)

    def acc_add_block(self, b):
        e, T = self.p.emit, self.vT[0]
        for r in range(4):
            tt = self.vt[r]
            e("v_accvgpr_read_b32", tt, self.acc[b][r])
            e("v_add_f32", tt, tt, T[r])
            e("v_accvgpr_write_b32", self.acc[b][r], tt)

    # ------------------------------------------------------------------ epilogue
    def c_addr_setup(self):
        """vC[n] = byte offset in C of D[4 q][r16] of block (0, n): row m0 + wm0 + 4 q, col n0 + wn0 + r16 + 16 n (an offset the bounds
        check always rejects for columns beyond N).  The rows of a lane -- + d within a block, + 16 per block row -- travel in the
        accesses' SCALAR offset, which gfx950 includes in a raw buffer's range check (scripts/probes/buffer_soffset.hip): rows beyond
        M fall off the end of srdC.  No vector address arithmetic between the stores."""
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        lane, r16, q = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, v(0))
        e("v_and_b32", r16, 15, lane)
        e("v_lshrrev_b32", q, 4, lane)
        e("s_add_u32", st[0], self.s_m0, self.s_wm0)
        e("v_lshl_add_u32", t[3], q, 2, st[0])
        e("v_mul_lo_u32", t[3], t[3], self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], r16)
        e("v_mul_lo_u32", t[6], t[4], self.s_csC4)
        e("v_add_u32", t[3], t[3], t[6])                 # row * ldc * 4 + col * csC * 4
        e("s_lshl_b32", st[1], self.s_csC4, 4)           # 16 columns further
        for n in range(c.TN):
            e("v_add_u32", t[5], 16 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            if n:
                e("v_add_u32", t[3], st[1], t[3])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[3], VCC)

    def c_rows(self, fn):
        """fn(i, d, soff): this lane's accumulator rows in C order; soff = the SGPR holding (16 i + d) * ldc * 4"""
        c, e, soff = self.c, self.p.emit, self.s_t[0]
        for i in range(c.TM):
            for d in range(4):
                if i == 0 and d == 0:
                    e("s_mov_b32", soff, 0)
                else:
                    e("s_mul_i32", soff, self.s_ldc4, 16 * i + d)
                fn(i, d, soff)

    def load_beta_c(self):
        """laser-order kernels, beta != 0: the running sum starts as beta * C0 (one rounding; gemm_ukernel_generic.nim:59-66, gemm.nim:158),
        loaded through the idle fragment registers and drained here, so the loop's counted waits (computed for beta == 0) stay correct"""
        c, p = self.c, self.p
        e = p.emit
        skip = p.label("nobeta")
        e("s_and_b32", self.s_t[1], self.s_beta, 0x7fffffff)
        e("s_cmp_eq_u32", self.s_t[1], 0)
        e("s_cbranch_scc1", skip)
        self.c_addr_setup()
        pool = [r[k] for slot in range(2) for r in (self.fa[slot] + self.fb[slot]) for k in range(4)]
        assert len(pool) >= 4 * c.TN
        for i in range(c.TM):
            for d in range(4):
                if i == 0 and d == 0:
                    e("s_mov_b32", self.s_t[0], 0)
                else:
                    e("s_mul_i32", self.s_t[0], self.s_ldc4, 16 * i + d)
                for n in range(c.TN):
                    # (runv: straight into the running sum's own registers)
                    dst = self.run[i * c.TN + n][d] if c.runv else pool[d * c.TN + n]
                    e("buffer_load_dword", dst, self.vC[n], self.srdC, self.s_t[0], offen=True)
            e("s_waitcnt", vmcnt=0)
            for d in range(4):
                for n in range(c.TN):
                    if c.runv:
                        x = self.run[i * c.TN + n][d]
                        e("v_mul_f32", x, self.s_beta, x)
                        continue
                    x = pool[d * c.TN + n]
                    e("v_mul_f32", x, self.s_beta, x)
                    e("v_accvgpr_write_b32", self.run[i * c.TN + n][d], x)
        p.place(skip)

    def fused_epilogue(self):
        """bias pointer != 0 or activation != 0: C = act((run | 0) + alpha * sum + bias), the bias added with one more rounding after the
        last slice, relu = max(x, 0) -- f32_kernel.Gen.fused_epilogue on the 16x16 lane map; a path of its own so that the plain epilogue
        carries none of it.  One-chain kernels: beta == 0 only (the launcher guarantees it).  Ends the program."""
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
        # bias descriptor: a null pointer gets a zero-size buffer (every load returns 0: x + 0 is x); bytes of the view =
        # ((M - 1) * rowStride + (N - 1) * colStride + 1) * 4 -- rows / columns of a ragged tile beyond it read 0
        e("s_and_b32", self.srdBias[1], self.srdBias[1], 0xffff)
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
        self.c_addr_setup()
        # bias offsets of D[4 q][r16] of block column n: (row * rsBias + col * csBias) * 4; the rows of a lane travel in the loads'
        # scalar offset like C's ((16 i + d) * rsBias * 4)
        lane, r16, q = t[0], t[1], t[2]
        vB = [self.vT[0][8 + n] for n in range(c.TN)]
        assert c.TN <= 8
        e("s_lshl_b32", st[2], self.s_epi[0], 2)          # rowStrideBias * 4
        e("s_add_u32", st[3], self.s_m0, self.s_wm0)
        e("v_lshl_add_u32", t[3], q, 2, st[3])
        e("v_mul_lo_u32", t[3], t[3], st[2])
        e("s_add_u32", st[3], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[3], r16)
        e("s_lshl_b32", st[4], self.s_epi[1], 2)          # colStrideBias * 4
        e("v_mul_lo_u32", t[4], t[4], st[4])
        e("v_add_u32", t[3], t[3], t[4])
        e("s_lshl_b32", st[4], st[4], 4)                  # 16 columns further
        for n in range(c.TN):
            if n == 0:
                e("v_mov_b32", vB[0], t[3])
            else:
                e("v_add_u32", vB[n], st[4], vB[n - 1])
        P = self.vT[0]
        soff, boff = st[0], st[1]
        for i in range(c.TM):
            for d in range(4):
                if i == 0 and d == 0:
                    e("s_mov_b32", soff, 0)
                    e("s_mov_b32", boff, 0)
                else:
                    e("s_mul_i32", soff, self.s_ldc4, 16 * i + d)
                    e("s_mul_i32", boff, st[2], 16 * i + d)
                for n in range(c.TN):
                    e("buffer_load_dword", P[n], vB[n], self.srdBias, boff, offen=True)
                e("s_waitcnt", vmcnt=0)
                assert c.TN <= 5      # (one temporary per block column: the whole row is computed, then the activation, then the stores)
                for n in range(c.TN):
                    b = i * c.TN + n
                    tt, uu = t[n], t[5 + n]
                    e("v_accvgpr_read_b32", tt, self.acc[b][d])
                    e("v_mul_f32", tt, self.s_alpha, tt)
                    if c.exact and c.runv:
                        e("v_add_f32", tt, self.run[b][d], tt)
                    elif c.exact:
                        e("v_accvgpr_read_b32", uu, self.run[b][d])
                        e("v_add_f32", tt, uu, tt)
                    e("v_add_f32", tt, P[n], tt)
                skip = p.label("norelu")
                e("s_cmp_lg_u32", self.s_epi[2], 1)
                e("s_cbranch_scc1", skip)
                for n in range(c.TN):
                    e("v_max_f32", t[n], 0, t[n])
                p.place(skip)
                for n in range(c.TN):
                    e("buffer_store_dword", t[n], self.vC[n], self.srdC, soff, offen=True)
        self.end_run()
        p.place(plain)

    def epilogue(self):
        """C = beta * C0 + alpha * (slice sums in order) -- gemm_ukernel_generic.nim:53-76 -- predicated by the descriptor's bounds check"""
        c, p = self.c, self.p
        e, t, st = p.emit, self.vt, self.s_t
        e("s_nop", 15)
        e("s_nop", 7)
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        self.vmq.clear()
        self.lgq.clear()
        self.mode_dispatch()
        self.fused_epilogue()
        self.c_addr_setup()
        if c.exact:
            # C = run + alpha * (the last slice's sum); run already carries beta * C0 and the earlier slices
            def row(i, d, soff):
                for n in range(c.TN):
                    b = i * c.TN + n
                    tt, uu = t[(2 * n) % 8], t[(2 * n + 1) % 8]
                    e("v_accvgpr_read_b32", tt, self.acc[b][d])
                    e("v_mul_f32", tt, self.s_alpha, tt)
                    if c.runv:
                        e("v_add_f32", tt, self.run[b][d], tt)
                    else:
                        e("v_accvgpr_read_b32", uu, self.run[b][d])
                        e("v_add_f32", tt, uu, tt)
                    if "cstores" not in c.ablate:
                        e("buffer_store_dword", tt, self.vC[n], self.srdC, soff, offen=True)
            self.c_rows(row)
        else:
            # one chain: C = beta * C0 + alpha * sum (beta == 0: C0 is never read, gemm_ukernel_generic.nim:59-66)
            withc, done = p.label("beta"), p.label("stored")
            e("s_and_b32", st[1], self.s_beta, 0x7fffffff)
            e("s_cmp_lg_u32", st[1], 0)
            e("s_cbranch_scc1", withc)

            def row0(i, d, soff):
                for n in range(c.TN):
                    tt = t[(2 * n) % 8]
                    e("v_accvgpr_read_b32", tt, self.acc[i * c.TN + n][d])
                    e("v_mul_f32", tt, self.s_alpha, tt)
                    e("buffer_store_dword", tt, self.vC[n], self.srdC, soff, offen=True)
            self.c_rows(row0)
            e("s_branch", done)
            p.place(withc)
            P = self.vT[0]

            def row1(i, d, soff):
                for n in range(c.TN):
                    e("buffer_load_dword", P[n], self.vC[n], self.srdC, soff, offen=True)
                e("s_waitcnt", vmcnt=0)
                for n in range(c.TN):
                    tt, x = t[4 + n % 4], P[n]
                    e("v_mul_f32", x, self.s_beta, x)
                    e("v_accvgpr_read_b32", tt, self.acc[i * c.TN + n][d])
                    e("v_mul_f32", tt, self.s_alpha, tt)
                    e("v_add_f32", tt, x, tt)
                    e("buffer_store_dword", tt, self.vC[n], self.srdC, soff, offen=True)
            self.c_rows(row1)
            p.place(done)
        self.end_run()


def make(name, **over):
    kw = dict(CONFIGS[name])
    kw.setdefault("pipe", True)       # pipelined tile transitions (the strided plan of the launcher): every tile of the family
    kw.update(over)
    return Gen16(Cfg(name, dtype="f32x16", **kw))


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
        sym = "lh_f32x16_" + name
        with open(os.path.join(args.out, sym + ".s"), "w") as f:
            f.write(kernel_text(g, sym))
        print(sym, len(g.p.ins), "instructions")
