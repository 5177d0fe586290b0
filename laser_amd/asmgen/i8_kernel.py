"""int32 / int64 GEMM on the int8 matrix cores, hand-scheduled: the limb-product kernel of gemm_i32_mfma.hip (arithmetic mod 2^32 over
balanced base-256 digits: 10 products of int8 planes per 32x32x32 block, four accumulator groups, one per power of 256) on the
generator's program structure -- one wave per SIMD, every accumulator in AGPRs (4 groups x 4 blocks x 16 = all 256), 3-stage
LDS ring with one barrier per K-tile, counted waits.

  v_mfma_i32_32x32x32_i8   32x32 block, 32 k per instruction; lane (lo = l % 32, hi = l / 32) feeds A[lo][16 hi .. 16 hi + 15] and
                           B[16 hi .. 16 hi + 15][lo] (16 bytes = 4 registers each); D as the f32 32x32 instruction.

Operands are the digit planes written by the packing pass (limb_planes.h, tile-major form): for every 128-row tile and 32-k
tile one contiguous 16-KiB block [plane p][k half h][row r][16 bytes] -- so the global->LDS stage is a lane-linear copy (thread t
moves bytes 16 t + 4096 i, fully coalesced, no address arithmetic) and a fragment read is 32 consecutive 16-byte chunks per half wave
(conflict-free).  Because the LDS image IS the memory image, the copy is an LDS-DMA: `buffer_load_dwordx4 voffset, srsrc, 0 offen lds`
writes lane l's 16 bytes at M0 + 16 l, M0 = stage + 4096 i + 1024 wave (one s_add_u32 + s_nop 0 per piece); tile t+2 is requested
during tile t into the stage tile t-1 was read from, and a counted vmcnt wait in front of tile t+1's barrier proves this wave's pieces
of tile t+1 have landed (the barrier covers the other waves').  No staging registers, no ds_write pass: +2 % on the kernel against the
register-staged loop (`dma=False`; profiles/r06/i8_dma_an.jsonl), same bits.  Workgroup tile 128x128, 2 x 2 waves of 64x64 (2 x 2 blocks); a K-tile is one
k-step: 40 MFMAs.  The fragments of tile t+1 (16 ds_read_b128) are read after the barrier of tile t into the other of two
register sets, so the loop is unrolled x6 (3 LDS stages x 2 fragment sets).  Any int32 alpha / beta (wrapping), K <= 8192 (no
fold of the accumulator groups: |G_s| <= 4 * 8192 * 2^14 = 2^29); everything else stays on the compiler-scheduled kernel.

int64 (`make("i64_64x64x32")`): eight planes, 36 products, eight accumulator groups = 128 AGPRs per 32x32 block, so a wave owns ONE
block and the workgroup tile is 64x64 (blocks of [8 planes][2 halves][64 rows][16 bytes] = the same 16 KiB); the epilogue sums
sext(G_s) << 8s in 64-bit (add-with-carry for s <= 3, shifted adds into the high word above), then alpha * (...) + beta * C0 wrapping."""
from .core import v, a, s, VCC, M0
from .f32_kernel import Gen, Cfg, kernel_text, KA_A, KA_LDA, KA_DBG, KA_SCHED, KA_SCHED2  # noqa: F401

KA_ALPHA64 = 72   # int64 alpha, beta (16 bytes): the slot of the f64 kernels' doubles

BLOCK = 16384      # bytes of one operand's (row tile, k tile) block: planes x 2 k halves x tile rows x 16 (int32: 4 x 2 x 128, int64: 8 x 2 x 64)


def products(np_):
    """the limb products (p, q) with p + q < np_ (what survives mod 256^np_), ordered so that consecutive MFMAs hit different
    accumulator groups s = p + q whenever another group still has products left"""
    left = {sg: [(p_, sg - p_) for p_ in range(sg + 1)] for sg in range(np_)}
    out, last = [], -1
    while any(left.values()):
        cands = sorted((sg for sg in left if left[sg] and sg != last), key=lambda sg: -len(left[sg])) or [sg for sg in left if left[sg]]
        sg = cands[0]
        out.append(left[sg].pop())
        last = sg
    return tuple(out)


class GenI8(Gen):
    def alloc(self):
        c, p = self.c, self.p
        S, V = p.salloc, p.valloc
        self.ka0 = S(8, align=4)
        self.ka1 = S(8, align=4)
        self.s_lda, self.s_ldb, self.s_ldc, self.s_M, self.s_N, self.s_K = (self.ka1[i] for i in range(6))
        self.s_alpha, self.s_beta = self.ka1[6], self.ka1[7]        # int32 (the float slots of the f32 kernels)
        self.srdA, self.srdB, self.srdC = S(4), S(4), S(4)
        self.s_rem = S()
        self.s_m0, self.s_n0, self.s_wave, self.s_wm0, self.s_wn0 = S(), S(), S(), S(), S()
        self.s_t = [S() for _ in range(6)]
        self.s_ldc4, self.s_ldc20 = S(), S()
        self.s_dma = S() if c.dma else None                      # LDS byte offset of this wave's 1-KiB share of a 4-KiB piece
        self.s_sc = S(8, align=4)                                # tile map constants (f32_kernel.py KA_SCHED)
        self.s_ab64 = S(4, align=4) if c.NP == 8 else None      # int64 alpha (lo, hi), beta (lo, hi): loaded by the epilogue
        self.acc = [[p.aalloc(16) for _ in range(c.WB * c.WB)] for _ in range(c.NP)]        # [power of 256][block = WB * i + n]
        self.fa = [[[V(4) for _ in range(c.NP)] for _ in range(c.WB)] for _ in range(2)]   # [set][i][plane]
        self.fb = [[[V(4) for _ in range(c.NP)] for _ in range(c.WB)] for _ in range(2)]   # [set][n][plane]
        self.stA = [V(4) for _ in range(4)] if not c.dma else None
        self.stB = [V(4) for _ in range(4)] if not c.dma else None
        self.vW = [V() for _ in range(c.NS)]
        self.RA = [V() for _ in range(c.NS)]
        self.RB = [V() for _ in range(c.NS)]
        self.vV = [V() for _ in range(4)]
        self.vC = [V() for _ in range(c.WB)]
        if c.debug:
            self.srdD = S(4)
            self.s_dslot = S()
            self.v_dbg = V()
        self.ndump = 0
        self.dump_names = []
        blk = V(12, align=4)
        self.vt = [blk[i] for i in range(10)]

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        c, p = self.c, self.p
        e, t, st = p.emit, self.vt, self.s_t
        p.note("int32 GEMM via int8 limb planes: 128x128 tile, 4 waves of 64x64, 40 MFMAs (10 limb products x 4 blocks) per 32-k tile")
        e("s_load_dwordx8", self.ka0, s(0, 2), KA_A)
        e("s_load_dwordx8", self.ka1, s(0, 2), KA_LDA)
        e("s_load_dwordx8", self.s_sc, s(0, 2), KA_SCHED)
        e("s_load_dword", st[5], s(0, 2), KA_SCHED2)
        if c.debug:
            e("s_load_dwordx2", self.srdD.sub(0, 2), s(0, 2), KA_DBG)
            e("s_waitcnt", lgkmcnt=0)
            e("s_and_b32", self.srdD[1], self.srdD[1], 0xffff)
            e("s_mov_b32", self.srdD[2], 0x10000000)
            e("s_mov_b32", self.srdD[3], 0x00020000)
            e("v_lshlrev_b32", self.v_dbg, 2, v(0))
        tid = v(0)
        lane, lo, hi = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, tid)
        e("v_lshrrev_b32", t[5], 6, tid)
        e("s_nop", 1, comment="VALU write -> v_readfirstlane of the same VGPR needs wait states")
        e("v_readfirstlane_b32", self.s_wave, t[5])
        e("s_nop", 3)
        e("v_and_b32", lo, 31, lane)
        e("v_lshrrev_b32", hi, 5, lane)
        e("s_lshr_b32", st[2], self.s_wave, 1)
        e("s_mul_i32", self.s_wm0, st[2], 32 * c.WB)
        e("s_and_b32", st[2], self.s_wave, 1)
        e("s_mul_i32", self.s_wn0, st[2], 32 * c.WB)
        # fragment reads: stage * STAGE [+ BLOCK for B] + plane * PLANE + hi * (PLANE / 2) + (w?0 + 32 blk + lo) * 16
        e("v_mul_u32_u24", t[5], c.PLANE // 2, hi)
        e("v_add_u32", t[6], self.s_wm0, lo)
        e("v_lshl_add_u32", t[6], t[6], 4, t[5])
        e("v_add_u32", t[7], self.s_wn0, lo)
        e("v_lshl_add_u32", t[7], t[7], 4, t[5])
        e("v_add_u32", t[7], BLOCK, t[7])
        for k in range(c.NS):
            e("v_add_u32", self.RA[k], k * c.STAGE, t[6])
            e("v_add_u32", self.RB[k], k * c.STAGE, t[7])
        e("v_lshlrev_b32", t[5], 4, tid)
        for k in range(c.NS):
            e("v_add_u32", self.vW[k], k * c.STAGE, t[5])
        for i in range(4):
            e("v_add_u32", self.vV[i], 4096 * i, t[5])
        # ---- tile coordinates, descriptors ----
        e("s_waitcnt", lgkmcnt=0)
        self.xcd_remap(st[4], s(2), self.s_sc[7], st[5], st[0])
        self.tile_coords(st[4], self.s_sc, st[0], st[1], (st[2], st[3], st[5]))
        e("s_mul_i32", self.s_m0, st[0], c.BM)
        e("s_mul_i32", self.s_n0, st[1], c.BN)
        A_, B_, C_ = self.ka0.sub(0, 2), self.ka0.sub(2, 2), self.ka0.sub(4, 2)
        e("s_lshl_b32", st[3], self.s_lda, 14, comment="bytes of one row tile's panel: k tiles * 16 KiB")
        for srd, base, pid in ((self.srdA, A_, st[0]), (self.srdB, B_, st[1])):
            e("s_mul_hi_u32", st[4], pid, st[3])
            e("s_mul_i32", st[2], pid, st[3])
            e("s_add_u32", srd[0], base[0], st[2])
            e("s_addc_u32", srd[1], base[1], st[4])
            e("s_and_b32", srd[1], srd[1], 0xffff)
            e("s_mov_b32", srd[2], st[3])
            e("s_mov_b32", srd[3], 0x00020000)
        e("s_lshl_b32", self.s_ldc4, self.s_ldc, c.ESH)
        e("s_mov_b32", self.srdC[0], C_[0])
        e("s_and_b32", self.srdC[1], C_[1], 0xffff)
        e("s_sub_u32", st[0], self.s_M, 1)
        e("s_mul_i32", st[0], st[0], self.s_ldc4)
        e("s_lshl_b32", st[2], self.s_N, c.ESH)
        e("s_add_u32", self.srdC[2], st[0], st[2])
        e("s_mov_b32", self.srdC[3], 0x00020000)
        e("s_mul_i32", self.s_ldc20, self.s_ldc4, 5)
        e("s_mov_b32", self.s_rem, self.s_lda)
        # ---- tile 0 -> LDS stage 0, tile 1 -> staging registers, fragments of tile 0 -> set 0 ----
        if c.dma:
            e("s_lshl_b32", self.s_dma, self.s_wave, 10)
            for k in range(c.NS - 1):
                self.run_ops(self.dma_ops(k))
                self.advance_srds()
        else:
            self.issue_loads_all()
            self.advance_srds()
            self.run_ops(self.store_ops(0))
            self.issue_loads_all()
            self.advance_srds()
        for sgrp in range(c.NP):
            for b in range(c.WB * c.WB):
                for r in range(16):
                    e("v_accvgpr_write_b32", self.acc[sgrp][b][r], 0)
        if c.dma:
            self.vm_wait(("D", 0, 7))
        self.lg_wait(None)
        e("s_barrier")
        self.run_ops(self.read_ops(0, 0))

    def issue_loads_all(self):
        for i in range(4):
            self.run_op(("loadA", i))
        for i in range(4):
            self.run_op(("loadB", i))

    def load_A_piece(self, pi):
        self.p.emit("buffer_load_dwordx4", self.stA[pi], self.vV[pi], self.srdA, 0, offen=True)
        self.vm_issue(("A", pi))

    def load_B_piece(self, pj):
        self.p.emit("buffer_load_dwordx4", self.stB[pj], self.vV[pj], self.srdB, 0, offen=True)
        self.vm_issue(("B", pj))

    def advance_srds(self, which=None):
        ops = []
        for srd in (self.srdA, self.srdB):
            ops += [("s_add_u32", srd[0], srd[0], BLOCK), ("s_addc_u32", srd[1], srd[1], 0),
                    ("s_sub_u32", srd[2], srd[2], BLOCK), ("s_cselect_b32", srd[2], 0, srd[2])]
        if "advance" in self.c.ablate:     # (timing experiment: every K-tile re-reads the first blocks -- cache-hot loads of real data)
            ops = []
        if which is None:
            for o in ops:
                self.p.emit(*o)
        return ops

    def dma_ops(self, k):
        """the 8 pieces of one K-tile's two blocks -> LDS stage k, no registers in between: piece i of a block lands at stage + 4096 i
        + 16 * thread = M0 + 16 * lane with M0 = stage + 4096 i + 1024 * wave (the lane-linear image the staged path writes)"""
        out = []
        for i in range(8):
            out.append([("ins", "s_add_u32", (M0, self.s_dma, k * self.c.STAGE + (BLOCK if i >= 4 else 0) + 4096 * (i % 4)), {}),
                        ("ins", "s_nop", (0,), {}), ("call", (lambda i=i, k=k: self.dma_piece(i, k)))])
        return [op for u in out for op in u]

    def dma_piece(self, i, k):
        self.p.emit("buffer_load_dwordx4", self.vV[i % 4], self.srdA if i < 4 else self.srdB, 0, offen=True, lds=True)
        self.vm_issue(("D", k, i))

    def store_ops(self, k):
        """staging registers -> LDS stage k: a lane-linear copy"""
        out = []
        for i in range(4):
            out += [("vmwait", ("A", i)), ("ldsw", "ds_write_b128", (self.vW[k], self.stA[i]), {"offset": 4096 * i})]
        for i in range(4):
            out += [("vmwait", ("B", i)), ("ldsw", "ds_write_b128", (self.vW[k], self.stB[i]), {"offset": BLOCK + 4096 * i})]
        return out

    def read_ops(self, k, fs):
        """the 16 fragments of the tile in LDS stage k -> register set fs, in the order the MFMAs want them"""
        c, ops = self.c, []
        for i in range(c.WB):
            for n in range(c.WB):
                if n == 0:
                    for pl in range(c.NP):
                        ops.append(("ldsr", "ds_read_b128", (self.fa[fs][i][pl], self.RA[k]), {"offset": c.PLANE * pl + 512 * i}, ("R", 0)))
                if i == 0:
                    for pl in range(c.NP):
                        ops.append(("ldsr", "ds_read_b128", (self.fb[fs][n][pl], self.RB[k]), {"offset": c.PLANE * pl + 512 * n}, ("R", 0)))
        return ops

    # ------------------------------------------------------------------ one K-tile: 40 MFMAs
    def tile_body(self, stage, fs):
        c, e = self.c, self.p.emit
        NMF = c.NMF
        gaps = {m: [] for m in range(-1, NMF)}
        nstage = (stage + 1) % c.NS
        # LDS stores of tile t+1 (staging registers -> stage t+1), each followed by the load of tile t+2 into the drained registers
        units = []
        if c.dma:
            d = self.dma_ops((stage + c.NS - 1) % c.NS)      # tile t+2 (t+3 with a four-stage ring) -> the stage tile t-1 was read from (every wave is past barrier t-1)
            units = [d[3 * i: 3 * i + 3] for i in range(8)]
        else:
            st = self.store_ops(nstage)
        for i in range(0 if c.dma else 8):
            units.append([st[2 * i], st[2 * i + 1]])
            if "looploads" not in c.ablate:       # (timing experiments: the loop keeps storing / multiplying the prologue's real data)
                units.append([("loadA", i) if i < 4 else ("loadB", i - 4)])
        if "loopreads" in c.ablate:
            fs = 0
        bar = c.bar_gap
        for k, u in enumerate(units):
            at = 1 + k * (bar - 2) // len(units)
            if c.dma and c.dma_sched == "early":        # (schedule probes, scripts/i8_probe.py: the spread placement is the shipped one)
                at = 1 + k
            elif c.dma and c.dma_sched == "late":
                at = bar - len(units) + k
            if c.dma and c.dma_sched == "split":        # M0 one gap ahead of its load: no wait state needed
                gaps[at - 1] += u[:1]
                gaps[at] += u[2:]
            else:
                gaps[at] += u
        if c.dma:
            gaps[bar].append(("vmwait", ("D", nstage, 7)))      # tile t+1 has landed (this wave's pieces; the barrier covers the others')
        gaps[bar].append(("barrier",))
        for k, o in enumerate(self.advance_srds(which="ops")):
            gaps[min(bar + 1 + k, NMF - 1)].append(("ins", o[0], o[1:], {}))
        # the 16 fragment reads of tile t+1 follow the barrier one per gap: all issued >= 8 gaps before the tile ends, so the
        # wait at the top of the next body finds them done
        rd = [] if "loopreads" in c.ablate else self.read_ops(nstage, 1 - fs)
        for k, o in enumerate(rd):
            gaps[min(bar + 1 + k * c.r_step, NMF - 2)].append(o)
        self.lg_wait({("R", 0)})
        m = 0
        for i in range(c.WB):
            for (pa, qb) in c.PRODUCTS:
                for n in range(c.WB):
                    acc = self.acc[pa + qb][c.WB * i + n]
                    e("v_mfma_i32_32x32x32_i8", acc, self.fa[fs][i][pa], self.fb[fs][n][qb], acc)
                    for op in gaps[m]:
                        self.run_op(op)
                    m += 1
        assert m == NMF

    def main_loop(self):
        c, p = self.c, self.p
        e = p.emit
        state0 = (list(self.vmq), list(self.lgq))
        L_done = p.label("done")
        nb = 6 if c.NS == 3 else 4             # bodies = lcm(LDS stages, fragment register sets)
        B = [p.label(f"tile_{k}") for k in range(nb)]
        e("raw", ".p2align 6")
        for k in range(nb):
            p.place(B[k])
            self.tile_body(k % c.NS, k % 2)
            assert (self.vmq, self.lgq) == state0 or c.ablate or (c.dma and len(self.vmq) == len(state0[0])), "loop-carried queue state differs"
            e("s_sub_u32", self.s_rem, self.s_rem, 1)
            e("s_cmp_eq_u32", self.s_rem, 0)
            e("s_cbranch_scc1", L_done)
            if k == nb - 1:
                e("s_branch", B[0])
        p.place(L_done)

    # ------------------------------------------------------------------ epilogue: G0 + (G1 << 8) + (G2 << 16) + (G3 << 24)
    def epilogue(self):
        if self.c.NP == 8:
            return self.epilogue64()
        c, p = self.c, self.p
        e, t = p.emit, self.vt
        e("s_nop", 15)
        e("s_nop", 7)
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        self.vmq.clear()
        self.lgq.clear()
        # C = beta * C0 + alpha * sum, all mod 2^32 (gemm_ukernel_generic.nim:53-76); beta == 0 never reads C
        withc, done = p.label("beta"), p.label("stored")
        e("s_cmp_lg_u32", self.s_beta, 0)
        e("s_cbranch_scc1", withc)
        for with_beta in (False, True):
            if with_beta:
                p.place(withc)
            self.c_addr_setup()
            for i in range(2):
                for q in range(4):
                    for rr in range(4):
                        r = 4 * q + rr
                        if with_beta:
                            for n in range(2):
                                e("buffer_load_dword", t[8 + n], self.vC[n], self.srdC, 0, offen=True)
                        for n in range(2):
                            b = 2 * i + n
                            x = [t[4 * n + j] for j in range(4)]
                            for j in range(4):
                                e("v_accvgpr_read_b32", x[j], self.acc[j][b][r])
                            e("v_lshl_add_u32", x[0], x[1], 8, x[0])
                            e("v_lshl_add_u32", x[0], x[2], 16, x[0])
                            e("v_lshl_add_u32", x[0], x[3], 24, x[0])
                            e("v_mul_lo_u32", x[0], x[0], self.s_alpha)
                        if with_beta:
                            e("s_waitcnt", vmcnt=0)
                            for n in range(2):
                                e("v_mul_lo_u32", t[8 + n], t[8 + n], self.s_beta)
                                e("v_add_u32", t[4 * n], t[4 * n], t[8 + n])
                        for n in range(2):
                            e("buffer_store_dword", t[4 * n], self.vC[n], self.srdC, 0, offen=True)
                        self.c_step(i, q, rr)
            if not with_beta:
                e("s_branch", done)
        p.place(done)
        e("s_endpgm")


    def c_addr_setup(self):
        if self.c.NP == 4:
            return Gen.c_addr_setup(self)
        # int64 elements: byte offset of (row m0 + wm0 + 4 hi, col n0 + wn0 + lo) with 8-byte elements
        c, e, t, st = self.c, self.p.emit, self.vt, self.s_t
        lane, lo, hi = t[0], t[1], t[2]
        e("v_and_b32", lane, 63, v(0))
        e("v_and_b32", lo, 31, lane)
        e("v_lshrrev_b32", hi, 5, lane)
        e("s_add_u32", st[0], self.s_m0, self.s_wm0)
        e("v_lshl_add_u32", t[3], hi, 2, st[0])
        e("v_mul_lo_u32", t[3], t[3], self.s_ldc4)
        e("s_add_u32", st[1], self.s_n0, self.s_wn0)
        e("v_add_u32", t[4], st[1], lo)
        e("v_lshl_add_u32", t[3], t[4], 3, t[3])
        for n in range(c.WB):
            e("v_add_u32", t[5], 32 * n, t[4])
            e("v_cmp_gt_u32", VCC, self.s_N, t[5])
            e("v_add_u32", t[6], 256 * n, t[3])
            e("v_mov_b32", t[7], 0x80000000)
            e("v_cndmask_b32", self.vC[n], t[7], t[6], VCC)

    def epilogue64(self):
        """C = beta * C0 + alpha * sum over s of sext(G_s) << 8s (mod 2^64; gemm_ukernel_generic.nim:53-76): 64-bit adds with carry
        for s <= 3, shifted adds into the high word above; alpha / beta are int64 at kernel-argument offset 72 / 80 (the doubles'
        slots of the f64 kernels).  Three bodies: alpha == 1 and beta == 0 (no multiplies), beta == 0 (never reads C), general."""
        c, p = self.c, self.p
        e, t, st = p.emit, self.vt, self.s_t
        e("s_nop", 15)
        e("s_nop", 7)
        e("s_waitcnt", vmcnt=0, lgkmcnt=0)
        self.vmq.clear()
        self.lgq.clear()
        ab = self.s_ab64
        e("s_load_dwordx4", ab, s(0, 2), KA_ALPHA64)
        e("s_waitcnt", lgkmcnt=0)
        L_mul, L_beta, L_done = p.label("alpha"), p.label("beta64"), p.label("stored64")
        e("s_xor_b32", st[2], ab[0], 1)
        e("s_or_b32", st[2], st[2], ab[1])
        e("s_or_b32", st[3], ab[2], ab[3])
        e("s_or_b32", st[2], st[2], st[3])
        e("s_cmp_lg_u32", st[2], 0)
        e("s_cbranch_scc1", L_mul)
        lo, hi, g, x = t[0], t[1], t[2], t[3]
        res = t[4:6]
        cc = t[6:8]
        for mode in ("plain", "alpha", "beta"):
            if mode == "alpha":
                p.place(L_mul)
                e("s_cmp_lg_u32", st[3], 0)
                e("s_cbranch_scc1", L_beta)
            elif mode == "beta":
                p.place(L_beta)
            self.c_addr_setup()
            for q in range(4):
                for rr in range(4):
                    r = 4 * q + rr
                    if mode == "beta":
                        e("buffer_load_dwordx2", v(cc[0].idx, 2), self.vC[0], self.srdC, 0, offen=True)
                    e("v_accvgpr_read_b32", lo, self.acc[0][0][r])
                    e("v_ashrrev_i32", hi, 31, lo)
                    for sg in range(1, 4):
                        e("v_accvgpr_read_b32", g, self.acc[sg][0][r])
                        e("v_lshlrev_b32", x, 8 * sg, g)
                        e("v_ashrrev_i32", g, 32 - 8 * sg, g)
                        e("v_add_co_u32", lo, VCC, lo, x)
                        e("v_addc_co_u32", hi, VCC, hi, g, VCC)
                    for sg in range(4, 8):
                        e("v_accvgpr_read_b32", g, self.acc[sg][0][r])
                        e("v_lshl_add_u32", hi, g, 8 * (sg - 4), hi)
                    if mode != "plain":
                        # (lo, hi) *= alpha mod 2^64: hi' = mulhi(lo, a_lo) + lo * a_hi + hi * a_lo
                        e("v_mul_hi_u32", x, lo, ab[0])
                        e("v_mul_lo_u32", g, lo, ab[1])
                        e("v_add_u32", x, x, g)
                        e("v_mul_lo_u32", g, hi, ab[0])
                        e("v_add_u32", hi, x, g)
                        e("v_mul_lo_u32", lo, lo, ab[0])
                    if mode == "beta":
                        e("s_waitcnt", vmcnt=0)
                        e("v_mul_hi_u32", x, cc[0], ab[2])
                        e("v_mul_lo_u32", g, cc[0], ab[3])
                        e("v_add_u32", x, x, g)
                        e("v_mul_lo_u32", g, cc[1], ab[2])
                        e("v_add_u32", x, x, g)
                        e("v_mul_lo_u32", g, cc[0], ab[2])
                        e("v_add_co_u32", lo, VCC, lo, g)
                        e("v_addc_co_u32", hi, VCC, hi, x, VCC)
                    e("v_mov_b32", res[0], lo)
                    e("v_mov_b32", res[1], hi)
                    e("buffer_store_dwordx2", v(res[0].idx, 2), self.vC[0], self.srdC, 0, offen=True)
                    self.c_step(0, q, rr)
            if mode != "beta":
                e("s_branch", L_done)
        p.place(L_done)
        e("s_endpgm")


def make(name="i32_128x128x32", **over):
    i64 = name.startswith("i64")
    kw = dict(BM=64 if i64 else 128, BN=64 if i64 else 128, BK=32, exact=False, bar_gap=18 if i64 else 20)   # (scripts/i8_probe.py: barrier 9 / 12 / 16 / 20 -> 275 / 279 / 283 / 286 Tint-op/s, int32)
    kw.update(over)
    dma = kw.pop("dma", True)                  # operand blocks straight into LDS (`buffer_load_dwordx4 ... lds`), no staging registers; False: the register-staged loop of rounds 4-5 (A/B, ablations)
    dma_sched = kw.pop("dma_sched", "spread")
    ns = kw.pop("stages", 3)                   # LDS ring: 3 stages (tile t+2 requested during tile t) or, LDS-DMA only, 4 (tile t+3)
    assert ns == 3 or (ns == 4 and dma)
    c = Cfg(name, persistent=False, **kw)       # one tile per workgroup (the limb kernels are never cut along K)
    c.dma = dma
    c.dma_sched = dma_sched
    c.NS = ns
    c.dtype = "i8"
    c.NP = 8 if i64 else 4                     # digit planes = bytes of the element
    c.WB = 1 if i64 else 2                     # 32x32 blocks per wave in each direction
    c.PRODUCTS = products(c.NP)
    c.PLANE = BLOCK // c.NP                    # bytes of one plane of a block: 2 k halves x tile rows x 16
    c.ESH = 3 if i64 else 2                    # log2(bytes of an element of C)
    c.TM = c.TN = c.WB
    c.NB, c.NMF, c.STAGE, c.NPA, c.NPB = c.WB * c.WB, c.WB * c.WB * len(c.PRODUCTS), 2 * BLOCK, 4, 4
    c.lds_bytes = c.lds_alloc = c.NS * c.STAGE
    return GenI8(c)


CONFIGS = {"i32_128x128x32": {}, "i64_64x64x32": {}}

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
        sym = "lh_" + name
        with open(os.path.join(args.out, sym + ".s"), "w") as f:
            f.write(kernel_text(g, sym))
        print(sym, len(g.p.ins), "instructions")
