"""CPU interpreter of an asmgen instruction list: one workgroup, wave64, gfx950 subset.

What it checks besides the values it computes:
  * every register read has its producing load retired by an s_waitcnt (loads are modelled as completing ONLY when a
    wait forces them -- the worst case the counted waits must be correct for);
  * no s_barrier is crossed with LDS writes still un-waited, and no two waves touch the same LDS dword in the same
    barrier epoch with at least one of them writing (cross-wave RAW / WAR / WAW races);
  * MFMA result -> VALU / v_accvgpr_read distance (software wait states), VALU-written SGPR -> VMEM distance;
  * LDS bank conflicts per instruction, by the lane-group / bank rules of MI355X_MICROARCH.md section LDS.
Values: 32-bit registers as uint32 numpy vectors over the 64 lanes; an f32 MFMA is evaluated as the k-ordered fmaf chain
it is on the hardware (products and sums through float64, then rounded -- exact for the integer-valued test data and the
same operation sequence as the reference model for random data).
"""
import numpy as np

from .core import Reg, Sym

U32 = np.uint32
LANES = 64

GROUPS_R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS_R128 += [[l + 32 for l in g] for g in GROUPS_R128]
GROUPS_32 = [list(range(0, 32)), list(range(32, 64))]
GROUPS_16 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
GROUPS_8 = [list(range(i, i + 8)) for i in range(0, 64, 8)]


class SimError(Exception):
    pass


def f32(x):
    return np.asarray(x, dtype=np.uint32).view(np.float32)


def u32(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def fma32(a, b, c):
    """fmaf on float32 vectors via float64 (product exact; one extra rounding only in halfway cases, applied equally
    in the reference model)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


# accumulator layouts as index maps [register][lane] -> (row, column) of the block (one fancy-indexed copy instead of a loop per register)
_L64 = np.arange(64)
_R32 = np.stack([(r & 3) + 8 * (r >> 2) + 4 * (_L64 >> 5) for r in range(16)])      # v_mfma_f32_32x32x2: element r of lane l
_C32 = np.broadcast_to(_L64 & 31, (16, 64)).copy()
_R16 = np.stack([4 * (_L64 >> 4) + d for d in range(4)])                            # v_mfma_f32_16x16x4: register d of lane l
_C16 = np.broadcast_to(_L64 & 15, (4, 64)).copy()


def fma32_outer(acol, brow, c):
    """c[i][j] = fmaf(a[i], b[j], c[i][j]) (fma32's arithmetic on the outer product, broadcast instead of materialised operands)"""
    return (acol.astype(np.float64)[:, None] * brow.astype(np.float64)[None, :] + c.astype(np.float64)).astype(np.float32)


class Memory:
    """Flat global memory: named numpy byte buffers at fake 64-bit addresses."""

    def __init__(self):
        self.bufs = []
        self.next = 0x7f0000000000

    def alloc(self, arr):
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.next += (len(raw) + 0xfff) // 0x1000 * 0x1000 + 0x10000
        self.bufs.append([base, raw])
        return base

    def _find(self, addr, n):
        for base, raw in self.bufs:
            if base <= addr and addr + n <= base + len(raw):
                return raw, addr - base
        raise SimError(f"global access outside every allocation: {addr:#x} + {n}")

    def read32(self, addr):
        raw, o = self._find(addr, 4)
        return int(raw[o:o + 4].view(np.uint32)[0])

    def read32_vec(self, addrs):
        """read32 of every address of an int64 array: one gather when they lie in ONE allocation (the usual case), else lane by lane
        (which also raises for an access outside every allocation)"""
        if len(addrs) == 0:
            return np.zeros(0, dtype=np.uint32)
        lo, hi = int(addrs.min()), int(addrs.max())
        for base, raw in self.bufs:
            if base <= lo and hi + 4 <= base + len(raw):
                off = addrs - base
                if len(raw) % 4 == 0 and not np.any(off & 3):
                    return raw.view(np.uint32)[off >> 2]
                break
        return np.array([self.read32(int(a_)) for a_ in addrs], dtype=np.uint32)

    def write32(self, addr, val):
        raw, o = self._find(addr, 4)
        raw[o:o + 4] = np.array([val], dtype=np.uint32).view(np.uint8)

    def get(self, base, dtype, shape):
        for b, raw in self.bufs:
            if b == base:
                return raw.view(dtype).reshape(shape)
        raise KeyError(base)


class Wave:
    def __init__(self, wg, wid):
        self.wg, self.wid = wg, wid
        self.v = np.zeros((256, LANES), dtype=U32)
        self.a = np.zeros((256, LANES), dtype=U32)
        self.s = np.zeros(104, dtype=np.uint64)  # 32-bit values (uint64 storage to make carries easy)
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.m0 = 0
        self.m0_wr = -10   # issue state of the last write of M0
        self.scc = 0
        self.pc = 0
        self.epoch = 0
        self.done = False
        self.vm = []    # in-order queue of outstanding VMEM ops: list of register-key lists
        self.lgkm = []  # in-order queue of outstanding LDS / SMEM ops: (kind, regkeys)
        self.pending = {}  # (kind, idx) -> what is loading it
        self.state = 0  # issue-state clock (MFMA = 16 states, s_nop n = n + 1, everything else 1)
        self.mfma_wr = {}  # agpr/vgpr idx key -> state at which an MFMA writing it was issued
        self.valu_sgpr_wr = {}  # sgpr idx -> state
        self.n_mfma = 0
        self.stats = {"bank_conflict_cycles": 0, "lds_ops": 0, "mfma": 0, "waits_vm": [], "ins": 0}

    def execmask(self):
        if getattr(self, "_em_exec", None) != self.exec:
            self._em_exec, self._em = self.exec, np.array([(self.exec >> l) & 1 for l in range(LANES)], dtype=bool)
        return self._em


class Workgroup:
    def __init__(self, prog, mem, kernarg_addr, wg_id=(0, 0), nwaves=4, lds_bytes=160 * 1024, check=True, bank_model=True):
        self.prog, self.mem = prog, mem
        self.bank_model = bank_model
        self.ins = [i for i in prog.ins]
        self.labels = {i.args[0]: n for n, i in enumerate(self.ins) if i.op == "label"}
        self.lds = np.zeros(lds_bytes // 4, dtype=U32)
        self.lds_w_epoch = np.full(lds_bytes // 4, -1, dtype=np.int32)
        self.lds_w_wave = np.full(lds_bytes // 4, -1, dtype=np.int32)
        self.lds_r_epoch = np.full((nwaves, lds_bytes // 4), -1, dtype=np.int32)
        self.check = check
        self.waves = []
        for w in range(nwaves):
            wv = Wave(self, w)
            wv.v[0] = np.arange(LANES, dtype=U32) + 64 * w
            wv.s[0] = kernarg_addr & 0xffffffff
            wv.s[1] = kernarg_addr >> 32
            wv.s[2], wv.s[3] = wg_id
            self.waves.append(wv)

    # ------------------------------------------------------------------ operand access
    def rd_s(self, w, x):
        if isinstance(x, Reg):
            assert x.kind == "s" and x.n == 1, x
            self._chk_pending(w, [("s", x.idx)])
            return int(w.s[x.idx]) & 0xffffffff
        if isinstance(x, Sym):
            if x.name == "m0":
                return w.m0
            raise SimError(f"32-bit scalar read of {x}")
        if isinstance(x, float):
            return int(np.array([x], dtype=np.float32).view(np.uint32)[0])
        return int(x) & 0xffffffff

    def rd_s64(self, w, x):
        if isinstance(x, Sym):
            return {"vcc": w.vcc, "exec": w.exec}[x.name]
        if isinstance(x, Reg):
            assert x.kind == "s" and x.n == 2
            self._chk_pending(w, x.regs())
            return (int(w.s[x.idx]) & 0xffffffff) | ((int(w.s[x.idx + 1]) & 0xffffffff) << 32)
        return int(x) & ((1 << 64) - 1)

    def wr_s64(self, w, x, val):
        val &= (1 << 64) - 1
        if isinstance(x, Sym):
            if x.name == "vcc":
                w.vcc = val
            elif x.name == "exec":
                w.exec = val
            else:
                raise SimError(x.name)
        else:
            w.s[x.idx] = val & 0xffffffff
            w.s[x.idx + 1] = val >> 32

    def rd_v(self, w, x):
        """32-bit source operand of a VALU op as a uint32 lane vector."""
        if isinstance(x, Reg):
            if x.kind == "v":
                self._chk_pending(w, [("v", x.idx)])
                self._chk_mfma_raw(w, ("v", x.idx))
                return w.v[x.idx].copy()
            if x.kind == "s":
                return np.full(LANES, self.rd_s(w, x), dtype=U32)
            if x.kind == "a":
                self._chk_pending(w, [("a", x.idx)])
                self._chk_mfma_raw(w, ("a", x.idx))
                return w.a[x.idx].copy()
        return np.full(LANES, self.rd_s(w, x), dtype=U32)

    def wr_v(self, w, x, val, masked=True):
        assert isinstance(x, Reg) and x.n == 1 and x.kind in ("v", "a"), x
        self._chk_waw(w, [(x.kind, x.idx)])
        arr = w.v if x.kind == "v" else w.a
        if masked and w.exec != (1 << 64) - 1:
            m = w.execmask()
            arr[x.idx][m] = np.asarray(val, dtype=U32)[m]
        else:
            arr[x.idx] = np.asarray(val, dtype=U32)

    def _chk_pending(self, w, keys):
        if not self.check:
            return
        for k in keys:
            if k in w.pending:
                raise SimError(f"wave {w.wid} pc {w.pc}: {self.ins[w.pc].text().strip()} reads {k[0]}{k[1]} while its load "
                               f"({w.pending[k]}) is not waited for")

    def _chk_waw(self, w, keys):
        if not self.check:
            return
        for k in keys:
            if k in w.pending:
                raise SimError(f"wave {w.wid} pc {w.pc}: {self.ins[w.pc].text().strip()} overwrites {k[0]}{k[1]} while a load "
                               f"into it ({w.pending[k]}) is outstanding")

    def _chk_mfma_raw(self, w, key):
        if self.check and key in w.mfma_wr and w.state - w.mfma_wr[key] < 19:
            raise SimError(f"wave {w.wid} pc {w.pc}: {self.ins[w.pc].text().strip()} reads {key[0]}{key[1]} only "
                           f"{w.state - w.mfma_wr[key]} wait states after the MFMA that writes it (need >= 19)")

    # ------------------------------------------------------------------ LDS helpers
    def _bank_cycles(self, addrs_per_lane, groups, nbanks, active):
        extra = 0
        for g in groups:
            per_bank = {}
            for l in g:
                if not active[l]:
                    continue
                for wd in addrs_per_lane[l]:
                    per_bank.setdefault(wd % nbanks, set()).add(wd)
            if per_bank:
                extra += max(len(x) for x in per_bank.values()) - 1
        return extra

    def _lds_access(self, w, byte_addrs, ndw, write, groups, nbanks):
        """byte_addrs: (LANES,) int64 start address per lane; ndw dwords each.  Race bookkeeping + bank model."""
        act = w.execmask()
        if np.any(byte_addrs[act] % 4):
            raise SimError(f"wave {w.wid} pc {w.pc}: unaligned LDS address")
        if ndw >= 2 and np.any(byte_addrs[act] % (8 if ndw == 2 else 16)) and self.ins[w.pc].op not in ("ds_write2_b32", "ds_read2_b32"):
            raise SimError(f"wave {w.wid} pc {w.pc}: {self.ins[w.pc].text().strip()}: LDS address not aligned to the access size")
        wd = (byte_addrs // 4).astype(np.int64)
        if np.any(wd[act] < 0) or np.any(wd[act] + ndw > len(self.lds)):
            raise SimError(f"wave {w.wid} pc {w.pc}: LDS address out of range")
        if self.check:
            for d in range(ndw):
                idx = wd[act] + d
                if write:
                    bad = (self.lds_w_epoch[idx] == w.epoch) & (self.lds_w_wave[idx] != w.wid)
                    if np.any(bad):
                        raise SimError(f"wave {w.wid} pc {w.pc} epoch {w.epoch}: LDS WAW race with wave {self.lds_w_wave[idx][bad][0]}")
                    for o in range(len(self.waves)):
                        if o != w.wid and np.any(self.lds_r_epoch[o][idx] == w.epoch):
                            raise SimError(f"wave {w.wid} pc {w.pc} epoch {w.epoch}: {self.ins[w.pc].text().strip()}: LDS WAR race: "
                                           f"wave {o} read these dwords in the same barrier epoch")
                else:
                    bad = (self.lds_w_epoch[idx] == w.epoch) & (self.lds_w_wave[idx] != w.wid)
                    if np.any(bad):
                        raise SimError(f"wave {w.wid} pc {w.pc} epoch {w.epoch}: {self.ins[w.pc].text().strip()}: LDS RAW race: "
                                       f"wave {self.lds_w_wave[idx][bad][0]} wrote these dwords in the same barrier epoch")
            for d in range(ndw):
                idx = wd[act] + d
                if write:
                    self.lds_w_epoch[idx] = w.epoch
                    self.lds_w_wave[idx] = w.wid
                else:
                    self.lds_r_epoch[w.wid][idx] = w.epoch
        if self.bank_model:      # (statistics only -- a quarter of the interpreter's time: the test suite runs without it)
            per_lane = [[int(wd[l]) + d for d in range(ndw)] for l in range(LANES)]
            w.stats["bank_conflict_cycles"] += self._bank_cycles(per_lane, groups, nbanks, act)
        w.stats["lds_ops"] += 1
        return wd

    # ------------------------------------------------------------------ wait counters
    def _retire_vm(self, w, n):
        while len(w.vm) > n:
            ent = w.vm.pop(0)
            if isinstance(ent, tuple) and ent[0] == "dma":      # an LDS-DMA piece lands when its wait retires it, not before:
                _, dw, vals = ent                               # a ds_read issued earlier sees the OLD bytes (MI355X_MICROARCH.md, item 7)
                self.lds[dw] = vals
                continue
            for k in ent:
                w.pending.pop(k, None)

    def _retire_lgkm(self, w, n):
        # SMEM returns out of order: with any SMEM outstanding only lgkmcnt(0) proves anything
        if any(kind == "smem" for kind, _ in w.lgkm) and n != 0:
            return
        while len(w.lgkm) > n:
            for k in w.lgkm.pop(0)[1]:
                w.pending.pop(k, None)

    # ------------------------------------------------------------------ buffer addressing
    def _srd(self, w, r):
        assert r.kind == "s" and r.n == 4
        self._chk_pending(w, r.regs())
        d = [int(w.s[r.idx + i]) & 0xffffffff for i in range(4)]
        base = d[0] | ((d[1] & 0xffff) << 32)
        stride = (d[1] >> 16) & 0x3fff
        assert stride == 0, "raw buffers only"
        return base, d[2]

    # ------------------------------------------------------------------ one instruction
    def step(self, w):
        i = self.ins[w.pc]
        op, A, M = i.op, i.args, i.mods
        nxt = w.pc + 1
        cost = 1
        if op in ("label", "comment", "raw"):
            w.pc = nxt
            return None
        w.stats["ins"] += 1
        rs, rv = self.rd_s, self.rd_v

        def sw(dst, val):
            if isinstance(dst, Sym):
                assert dst.name == "m0", dst
                w.m0, w.m0_wr = int(val) & 0xffffffff, w.state
                return
            assert dst.kind == "s" and dst.n == 1
            w.s[dst.idx] = int(val) & 0xffffffff
            w.valu_sgpr_wr.pop(dst.idx, None)

        def to_i32(x):
            return x - (1 << 32) if x & 0x80000000 else x

        if op == "s_mov_b32":
            sw(A[0], rs(w, A[1]))
        elif op == "s_mov_b64":
            self.wr_s64(w, A[0], self.rd_s64(w, A[1]))
        elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_add_i32", "s_sub_i32"):
            x, y = rs(w, A[1]), rs(w, A[2])
            if op == "s_add_u32" or op == "s_add_i32":
                r = x + y
                sw(A[0], r)
                w.scc = int(r >> 32) if op == "s_add_u32" else 0
            elif op == "s_addc_u32":
                r = x + y + w.scc
                sw(A[0], r)
                w.scc = int(r >> 32)
            else:
                r = x - y
                sw(A[0], r)
                w.scc = int(y > x) if op == "s_sub_u32" else 0
        elif op == "s_mul_i32":
            sw(A[0], rs(w, A[1]) * rs(w, A[2]))
        elif op == "s_mul_hi_u32":
            sw(A[0], (rs(w, A[1]) * rs(w, A[2])) >> 32)
        elif op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32", "s_xor_b32", "s_andn2_b32"):
            x, y = rs(w, A[1]), rs(w, A[2])
            r = {"s_lshl_b32": x << (y & 31), "s_lshr_b32": x >> (y & 31), "s_and_b32": x & y, "s_or_b32": x | y,
                 "s_xor_b32": x ^ y, "s_andn2_b32": x & ~y}[op] & 0xffffffff
            sw(A[0], r)
            w.scc = int(r != 0)
        elif op == "s_bfe_u32":
            x, y = rs(w, A[1]), rs(w, A[2])
            off, width = y & 31, (y >> 16) & 0x7f
            r = (x >> off) & ((1 << width) - 1) if width else 0
            sw(A[0], r)
            w.scc = int(r != 0)
        elif op == "s_subb_u32":
            x, y = rs(w, A[1]), rs(w, A[2])
            r = x - y - w.scc
            sw(A[0], r)
            w.scc = int(y + w.scc > x)
        elif op in ("s_and_b64", "s_or_b64", "s_andn2_b64"):
            x, y = self.rd_s64(w, A[1]), self.rd_s64(w, A[2])
            r = {"s_and_b64": x & y, "s_or_b64": x | y, "s_andn2_b64": x & ~y}[op] & ((1 << 64) - 1)
            self.wr_s64(w, A[0], r)
            w.scc = int(r != 0)
        elif op == "s_cselect_b64":
            self.wr_s64(w, A[0], self.rd_s64(w, A[1]) if w.scc else self.rd_s64(w, A[2]))
        elif op in ("s_min_u32", "s_max_u32"):
            x, y = rs(w, A[1]), rs(w, A[2])
            r = min(x, y) if op == "s_min_u32" else max(x, y)
            sw(A[0], r)
            w.scc = int(r == x)
        elif op.startswith("s_cmp_"):
            x, y = rs(w, A[0]), rs(w, A[1])
            if op.endswith("_i32"):
                x, y = to_i32(x), to_i32(y)
            rel = op.split("_")[2]
            w.scc = int({"eq": x == y, "lg": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[rel])
        elif op == "s_bitcmp1_b32":
            w.scc = (rs(w, A[0]) >> (rs(w, A[1]) & 31)) & 1
        elif op == "s_cselect_b32":
            sw(A[0], rs(w, A[1]) if w.scc else rs(w, A[2]))
        elif op in ("s_cbranch_scc0", "s_cbranch_scc1", "s_branch"):
            take = op == "s_branch" or (w.scc == (1 if op.endswith("1") else 0))
            if take:
                nxt = self.labels[A[0] if isinstance(A[0], str) else A[0].name]
        elif op == "s_endpgm":
            w.done = True
            w.pc = nxt
            return "end"
        elif op == "s_nop":
            cost = int(A[0]) + 1
        elif op == "s_sleep":
            # only ever executed inside a spin on a flag another workgroup sets: the interpreter runs workgroups one after the
            # other (producers first), so a flag that is not set yet never will be
            raise SimError(f"wave {w.wid} pc {w.pc}: spinning on a flag that no earlier workgroup has set (wrong run order, or a lost release)")
        elif op == "buffer_wbl2":
            w.vm.append([])
        elif op == "buffer_inv":
            pass
        elif op == "s_barrier":
            if self.check and any(kind == "ldsw" for kind, _ in w.lgkm):
                raise SimError(f"wave {w.wid} pc {w.pc}: s_barrier crossed with LDS writes not waited for")
            w.pc = nxt
            w.state += 1
            return "barrier"
        elif op == "s_waitcnt":
            if "vmcnt" in M:
                w.stats["waits_vm"].append((w.pc, len(w.vm), M["vmcnt"]))
                self._retire_vm(w, M["vmcnt"])
            if "lgkmcnt" in M:
                self._retire_lgkm(w, M["lgkmcnt"])
        elif op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8"):
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8}[op]
            base = self.rd_s64(w, A[1])
            off = rs(w, A[2])
            assert A[0].n == n
            for k in range(n):
                w.s[A[0].idx + k] = self.mem.read32(base + off + 4 * k)
                w.pending[("s", A[0].idx + k)] = "s_load"
            w.lgkm.append(("smem", A[0].regs()))
        elif op == "s_memtime":
            self.wr_s64(w, A[0], w.state * 4)
            w.lgkm.append(("smem", []))
        # ------------------------------------------------------------ VALU
        elif op == "v_mov_b32":
            self.wr_v(w, A[0], rv(w, A[1]))
        elif op in ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_and_b32", "v_or_b32", "v_xor_b32",
                    "v_lshlrev_b32", "v_lshrrev_b32", "v_min_u32", "v_max_u32", "v_mul_hi_u32"):
            x, y = rv(w, A[1]).astype(np.uint64), rv(w, A[2]).astype(np.uint64)
            r = {"v_add_u32": lambda: x + y, "v_sub_u32": lambda: x - y, "v_subrev_u32": lambda: y - x,
                 "v_mul_lo_u32": lambda: x * y, "v_mul_hi_u32": lambda: (x * y) >> np.uint64(32), "v_mul_u32_u24": lambda: (x & 0xffffff) * (y & 0xffffff),
                 "v_and_b32": lambda: x & y, "v_or_b32": lambda: x | y, "v_xor_b32": lambda: x ^ y,
                 "v_lshlrev_b32": lambda: y << (x & 31), "v_lshrrev_b32": lambda: y >> (x & 31),
                 "v_min_u32": lambda: np.minimum(x, y), "v_max_u32": lambda: np.maximum(x, y)}[op]()
            self.wr_v(w, A[0], (r & 0xffffffff).astype(U32))
        elif op in ("v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_lshl_or_b32", "v_mad_u32_u24", "v_bfe_u32", "v_add3_u32", "v_xad_u32"):
            x, y, z = (rv(w, A[k]).astype(np.uint64) for k in (1, 2, 3))
            r = {"v_lshl_add_u32": lambda: (x << (y & 31)) + z, "v_add_lshl_u32": lambda: (x + y) << (z & 31),
                 "v_and_or_b32": lambda: (x & y) | z, "v_lshl_or_b32": lambda: (x << (y & 31)) | z,
                 "v_mad_u32_u24": lambda: (x & 0xffffff) * (y & 0xffffff) + z,
                 "v_bfe_u32": lambda: (x >> (y & 31)) & ((np.uint64(1) << (z & 31)) - np.uint64(1)),
                 "v_add3_u32": lambda: x + y + z, "v_xad_u32": lambda: (x ^ y) + z}[op]()
            self.wr_v(w, A[0], (r & 0xffffffff).astype(U32))
        elif op == "v_ashrrev_i32":
            x = rv(w, A[1]).astype(np.int64) & 31
            y = rv(w, A[2]).astype(np.uint32).view(np.int32).astype(np.int64)
            self.wr_v(w, A[0], ((y >> x) & 0xffffffff).astype(U32))
        elif op in ("v_add_co_u32", "v_addc_co_u32"):
            # D, carry-out (vcc or an SGPR pair), S0, S1 [, carry-in]
            x, y = rv(w, A[2]).astype(np.uint64), rv(w, A[3]).astype(np.uint64)
            cin = np.zeros(LANES, dtype=np.uint64)
            if op == "v_addc_co_u32":
                mk = self.rd_s64(w, A[4])
                cin = np.array([(mk >> l) & 1 for l in range(LANES)], dtype=np.uint64)
            r = x + y + cin
            cout = 0
            act = w.execmask()
            for l in range(LANES):
                if act[l] and int(r[l]) >> 32:
                    cout |= 1 << l
            self.wr_v(w, A[0], (r & np.uint64(0xffffffff)).astype(U32))
            self.wr_s64(w, A[1], cout)
        elif op == "v_readfirstlane_b32":
            k_ = (A[1].kind, A[1].idx)
            if self.check and k_ in w.valu_wr_state and w.state - w.valu_wr_state[k_] < 3:
                raise SimError(f"wave {w.wid} pc {w.pc}: v_readfirstlane reads {A[1]} {w.state - w.valu_wr_state[k_]} states after a VALU "
                               f"wrote it (hardware returned the OLD value without wait states)")
            first = next(l for l in range(LANES) if (w.exec >> l) & 1)
            sw(A[0], int(rv(w, A[1])[first]))
            w.valu_sgpr_wr[A[0].idx] = w.state
        elif op == "v_swap_b32":
            x, y = rv(w, A[0]), rv(w, A[1])
            self.wr_v(w, A[0], y)
            self.wr_v(w, A[1], x)
        elif op == "v_accvgpr_read_b32":
            assert A[0].kind == "v" and A[1].kind == "a"
            self.wr_v(w, A[0], rv(w, A[1]))
        elif op == "v_accvgpr_write_b32":
            assert A[0].kind == "a"
            self.wr_v(w, A[0], rv(w, A[1]))
        elif op in ("v_add_f32", "v_mul_f32", "v_sub_f32", "v_max_f32"):
            x, y = f32(rv(w, A[1])), f32(rv(w, A[2]))
            with np.errstate(all="ignore"):
                r = {"v_add_f32": x + y, "v_mul_f32": x * y, "v_sub_f32": x - y, "v_max_f32": np.fmax(x, y)}[op]
            self.wr_v(w, A[0], u32(r.astype(np.float32)))
        elif op == "v_mov_b64":
            assert A[0].n == 2 and A[0].kind == "v"
            if isinstance(A[1], Reg):
                assert A[1].n == 2
                for h in range(2):
                    self.wr_v(w, A[0][h], rv(w, A[1][h]))
            else:
                val = int(A[1])
                assert 0 <= val <= 64, "inline constant"
                self.wr_v(w, A[0][0], np.full(LANES, val, dtype=U32))
                self.wr_v(w, A[0][1], np.zeros(LANES, dtype=U32))
        elif op == "v_pk_add_f32":
            assert A[0].n == 2 and A[1].n == 2 and A[2].n == 2
            for h in range(2):
                x, y = f32(rv(w, A[1][h])), f32(rv(w, A[2][h]))
                with np.errstate(all="ignore"):
                    self.wr_v(w, A[0][h], u32((x + y).astype(np.float32)))
        elif op.startswith("v_cmp_"):
            rel, ty = op.split("_")[2], op.split("_")[3]
            x, y = rv(w, A[1]), rv(w, A[2])
            if ty == "i32":
                x, y = x.view(np.int32), y.view(np.int32)
            elif ty == "f32":
                x, y = f32(x), f32(y)
            r = {"eq": x == y, "ne": x != y, "lg": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[rel]
            m = w.execmask()
            val = 0
            for l in range(LANES):
                if r[l] and m[l]:
                    val |= 1 << l
            self.wr_s64(w, A[0], val)
            if isinstance(A[0], Reg):
                w.valu_sgpr_wr[A[0].idx] = w.valu_sgpr_wr[A[0].idx + 1] = w.state
        elif op == "v_cndmask_b32":
            x, y = rv(w, A[1]), rv(w, A[2])
            mk = self.rd_s64(w, A[3])
            sel = np.array([(mk >> l) & 1 for l in range(LANES)], dtype=bool)
            self.wr_v(w, A[0], np.where(sel, y, x))
        elif op == "v_mfma_f32_32x32x2_f32":
            cost = 16
            D, SA, SB, SC = A
            assert D.n == 16 and SA.n == 1 and SB.n == 1
            if self.check:
                for r_ in (SA, SB):
                    k = (r_.kind, r_.idx)
                    if k in w.valu_wr_state and w.state - w.valu_wr_state[k] < 2:
                        raise SimError(f"wave {w.wid} pc {w.pc}: MFMA reads {r_} right after a VALU wrote it")
            av = f32(rv(w, SA))
            bv = f32(rv(w, SB))
            if isinstance(SC, Reg):
                assert SC.n == 16
                self._chk_pending(w, SC.regs())
                arr = w.v if SC.kind == "v" else w.a
                if self.check:
                    for kk in SC.regs():
                        if kk in w.mfma_wr and w.state - w.mfma_wr[kk] < 16 and not (SC.kind == D.kind and SC.idx == D.idx):
                            raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC overlaps a different in-flight MFMA result")
                        if kk in w.valu_wr_state and w.state - w.valu_wr_state[kk] < 3:
                            raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC {SC} read {w.state - w.valu_wr_state[kk]} states after a VALU wrote it")
                cm = f32(arr[SC.idx:SC.idx + 16].copy())  # [r][lane]
            else:
                cm = np.full((16, LANES), f32(np.array([self.rd_s(w, SC)], dtype=U32))[0], dtype=np.float32)
            # C tile [32][32]: element r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l & 31
            Ct = np.zeros((32, 32), dtype=np.float32)
            Ct[_R32, _C32] = cm
            for kk in range(2):
                # A[i][k]: lane i + 32k; B[k][j]: lane j + 32k
                Ct = fma32_outer(av[32 * kk:32 * kk + 32], bv[32 * kk:32 * kk + 32], Ct)
            arr = w.v if D.kind == "v" else w.a
            self._chk_waw(w, D.regs())
            arr[D.idx:D.idx + 16] = u32(Ct[_R32, _C32])
            for r in range(16):
                w.mfma_wr[(D.kind, D.idx + r)] = w.state
            w.n_mfma += 1
            w.stats["mfma"] += 1
        elif op == "v_mfma_f32_16x16x4_f32":
            # D[i][j] += fmaf chain over k = 0..3 (ascending) of A[i][k] * B[k][j]; lane l holds A[l % 16][l / 16], B[l / 16][l % 16] and
            # D[4 * (l / 16) + d][l % 16] for d = 0..3 (one register each); 8 passes, 40 cycles until a dependent MFMA
            cost = 8
            D, SA, SB, SC = A
            assert D.n == 4 and SA.n == 1 and SB.n == 1
            if self.check:
                for r_ in (SA, SB):
                    k = (r_.kind, r_.idx)
                    if k in w.valu_wr_state and w.state - w.valu_wr_state[k] < 2:
                        raise SimError(f"wave {w.wid} pc {w.pc}: MFMA reads {r_} right after a VALU wrote it")
            av = f32(rv(w, SA))
            bv = f32(rv(w, SB))
            lanes = np.arange(LANES)
            Ct = np.zeros((16, 16), dtype=np.float32)
            if isinstance(SC, Reg):
                assert SC.n == 4
                self._chk_pending(w, SC.regs())
                arr = w.v if SC.kind == "v" else w.a
                if self.check:
                    for kk in SC.regs():
                        if kk in w.mfma_wr and w.state - w.mfma_wr[kk] < 10 and not (SC.kind == D.kind and SC.idx == D.idx):
                            raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC overlaps a different in-flight MFMA result")
                        if SC.kind == D.kind and SC.idx == D.idx and kk in w.mfma_wr and w.state - w.mfma_wr[kk] < 8:
                            raise SimError(f"wave {w.wid} pc {w.pc}: back-to-back dependent 16x16x4 MFMAs on {D} (40-cycle dependent latency: the matrix pipe would idle)")
                        if kk in w.valu_wr_state and w.state - w.valu_wr_state[kk] < 3:
                            raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC {SC} read {w.state - w.valu_wr_state[kk]} states after a VALU wrote it")
                cm = f32(arr[SC.idx:SC.idx + 4].copy())  # [d][lane]
                Ct[_R16, _C16] = cm
            else:
                assert int(SC) == 0
            for kk in range(4):
                # A[i][k]: lane i + 16k; B[k][j]: lane j + 16k
                Ct = fma32_outer(av[16 * kk:16 * kk + 16], bv[16 * kk:16 * kk + 16], Ct)
            arr = w.v if D.kind == "v" else w.a
            self._chk_waw(w, D.regs())
            arr[D.idx:D.idx + 4] = u32(Ct[_R16, _C16])
            for d in range(4):
                w.mfma_wr[(D.kind, D.idx + d)] = w.state
            w.n_mfma += 1
            w.stats["mfma"] += 1
        elif op == "v_mfma_i32_32x32x32_i8":
            # D[i][j] += sum over 32 k of int8 A[i][k] * int8 B[k][j] (int32, wrapping); lane (lo, hi) holds A[lo][16 hi + 0..15]
            # and B[16 hi + 0..15][lo] as 4 registers each; D as the f32 32x32 instruction
            cost = 8
            D, SA, SB, SC = A
            assert D.n == 16 and SA.n == 4 and SB.n == 4 and isinstance(SC, Reg) and SC.n == 16
            if self.check:
                for kk in SC.regs():
                    if kk in w.mfma_wr and w.state - w.mfma_wr[kk] < 8 and not (SC.kind == D.kind and SC.idx == D.idx):
                        raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC overlaps a different in-flight MFMA result")
                    if kk in w.valu_wr_state and w.state - w.valu_wr_state[kk] < 3:
                        raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC {SC} read right after a VALU wrote it")

            def bytes_of(R):
                self._chk_pending(w, R.regs())
                arr_ = w.v if R.kind == "v" else w.a
                words = np.stack([arr_[R.idx + d] for d in range(4)], axis=1).astype(np.uint32)   # [lane][4]
                return words.view(np.int8).reshape(LANES, 16).astype(np.int64)                     # [lane][16 k]
            ab, bb = bytes_of(SA), bytes_of(SB)
            lanes = np.arange(LANES)
            Amat = np.zeros((32, 32), dtype=np.int64)
            Bmat = np.zeros((32, 32), dtype=np.int64)
            for l in range(LANES):
                Amat[l & 31, 16 * (l >> 5):16 * (l >> 5) + 16] = ab[l]
                Bmat[16 * (l >> 5):16 * (l >> 5) + 16, l & 31] = bb[l]
            arr = w.a if SC.kind == "a" else w.v
            Ct = np.zeros((32, 32), dtype=np.int64)
            for r in range(16):
                Ct[(r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5), lanes & 31] = arr[SC.idx + r].astype(np.int64)
            Ct = (Ct + Amat @ Bmat) & 0xffffffff
            arr = w.v if D.kind == "v" else w.a
            self._chk_waw(w, D.regs())
            for r in range(16):
                arr[D.idx + r] = Ct[(r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5), lanes & 31].astype(U32)
                w.mfma_wr[(D.kind, D.idx + r)] = w.state
            w.n_mfma += 1
            w.stats["mfma"] += 1
        elif op == "v_mfma_f64_16x16x4_f64":
            # D[i][j] += sum over k = 0..3 (ascending, fused) of A[i][k] * B[k][j]; lane l holds A[l % 16][l / 16], B[l / 16][l % 16],
            # and D[l / 16 + 4 * d][l % 16] for d = 0..3 (two registers each).  The products and sums are done in float64
            # WITHOUT an exact fused multiply-add model: the f64 tests use integer-valued operands, for which every product
            # and partial sum is exact (fma == mul + add); the accumulation ORDER is validated on hardware.
            cost = 16
            D, SA, SB, SC = A
            assert D.n == 8 and SA.n == 2 and SB.n == 2
            if self.check:
                for r_ in list(SA.regs()) + list(SB.regs()):
                    if r_ in w.valu_wr_state and w.state - w.valu_wr_state[r_] < 2:
                        raise SimError(f"wave {w.wid} pc {w.pc}: MFMA reads {r_} right after a VALU wrote it")

            def rd64(R):
                arr_ = w.v if R.kind == "v" else w.a
                self._chk_pending(w, R.regs())
                lo = arr_[R.idx].astype(np.uint64)
                hi = arr_[R.idx + 1].astype(np.uint64)
                return (lo | (hi << np.uint64(32))).view(np.float64)
            av, bv = rd64(SA), rd64(SB)
            lanes = np.arange(LANES)
            Ct = np.zeros((16, 16), dtype=np.float64)
            if isinstance(SC, Reg):
                assert SC.n == 8
                if self.check:
                    for kk in SC.regs():
                        if kk in w.mfma_wr and w.state - w.mfma_wr[kk] < 16 and not (SC.kind == D.kind and SC.idx == D.idx):
                            raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC overlaps a different in-flight MFMA result")
                        if kk in w.valu_wr_state and w.state - w.valu_wr_state[kk] < 3:
                            raise SimError(f"wave {w.wid} pc {w.pc}: MFMA srcC {SC} read {w.state - w.valu_wr_state[kk]} states after a VALU wrote it")
                for d in range(4):
                    Ct[(lanes >> 4) + 4 * d, lanes & 15] = rd64(SC.sub(2 * d, 2))
            else:
                assert int(SC) == 0
            for kk in range(4):
                arow = av[16 * kk:16 * kk + 16]
                bcol = bv[16 * kk:16 * kk + 16]
                with np.errstate(all="ignore"):
                    Ct = arow[:, None] * bcol[None, :] + Ct
            arr = w.v if D.kind == "v" else w.a
            self._chk_waw(w, D.regs())
            for d in range(4):
                bits = Ct[(lanes >> 4) + 4 * d, lanes & 15].copy().view(np.uint64)
                arr[D.idx + 2 * d] = (bits & np.uint64(0xffffffff)).astype(U32)
                arr[D.idx + 2 * d + 1] = (bits >> np.uint64(32)).astype(U32)
                w.mfma_wr[(D.kind, D.idx + 2 * d)] = w.state
                w.mfma_wr[(D.kind, D.idx + 2 * d + 1)] = w.state
            w.n_mfma += 1
            w.stats["mfma"] += 1
        elif op in ("v_add_f64", "v_mul_f64"):
            assert A[0].n == 2

            def rd64v(X):
                if isinstance(X, Reg) and X.kind == "v":
                    assert X.n == 2
                    lo, hi = rv(w, X[0]).astype(np.uint64), rv(w, X[1]).astype(np.uint64)
                    return (lo | (hi << np.uint64(32))).view(np.float64)
                if isinstance(X, Reg) and X.kind == "s":
                    assert X.n == 2
                    return np.full(LANES, np.array([self.rd_s64(w, X)], dtype=np.uint64).view(np.float64)[0])
                raise SimError(f"f64 operand {X}")
            x, y = rd64v(A[1]), rd64v(A[2])
            with np.errstate(all="ignore"):
                r = (x + y) if op == "v_add_f64" else (x * y)
            bits = r.view(np.uint64)
            self.wr_v(w, A[0][0], (bits & np.uint64(0xffffffff)).astype(U32))
            self.wr_v(w, A[0][1], (bits >> np.uint64(32)).astype(U32))
        # ------------------------------------------------------------ LDS
        elif op in ("ds_read_b128", "ds_read_b64", "ds_read_b32"):
            ndw = {"ds_read_b128": 4, "ds_read_b64": 2, "ds_read_b32": 1}[op]
            addr = rv(w, A[1]).astype(np.int64) + int(M.get("offset", 0))
            groups, nb = {4: (GROUPS_R128, 64), 2: (GROUPS_32, 64), 1: (GROUPS_32, 32)}[ndw]
            wd = self._lds_access(w, addr, ndw, False, groups, nb)
            assert A[0].n == ndw
            self._chk_waw(w, A[0].regs())
            act = w.execmask()
            for d in range(ndw):
                dst = w.v if A[0].kind == "v" else w.a
                dst[A[0].idx + d][act] = self.lds[wd[act] + d]
                w.pending[(A[0].kind, A[0].idx + d)] = op
            w.lgkm.append(("ldsr", A[0].regs()))
        elif op == "ds_read_addtid_b32":
            # LDS address = M0[15:0] + offset + 4 * lane (no address register); S_MOV to M0 -> add-TID LDS op needs one wait state
            if self.check and w.state - w.m0_wr < 2:
                raise SimError(f"wave {w.wid} pc {w.pc}: ds_read_addtid right after M0 was written (needs s_nop 0)")
            addr = (w.m0 & 0xffff) + int(M.get("offset", 0)) + 4 * np.arange(LANES, dtype=np.int64)
            wd = self._lds_access(w, addr, 1, False, GROUPS_32, 32)
            assert A[0].n == 1 and A[0].kind == "v"
            self._chk_waw(w, A[0].regs())
            act = w.execmask()
            w.v[A[0].idx][act] = self.lds[wd[act]]
            w.pending[("v", A[0].idx)] = op
            w.lgkm.append(("ldsr", A[0].regs()))
        elif op in ("ds_write_b32", "ds_write_b64", "ds_write_b128"):
            ndw = {"ds_write_b32": 1, "ds_write_b64": 2, "ds_write_b128": 4}[op]
            addr = rv(w, A[0]).astype(np.int64) + int(M.get("offset", 0))
            groups = {1: GROUPS_32, 2: GROUPS_16, 4: GROUPS_8}[ndw]
            wd = self._lds_access(w, addr, ndw, True, groups, 32)
            assert A[1].n == ndw
            act = w.execmask()
            for d in range(ndw):
                self.lds[wd[act] + d] = rv(w, A[1][d])[act]
            w.lgkm.append(("ldsw", []))
        elif op == "ds_write2_b64":
            base = rv(w, A[0]).astype(np.int64)
            act = w.execmask()
            for which, data in ((0, A[1]), (1, A[2])):
                assert data.n == 2
                addr = base + 8 * int(M.get(f"offset{which}", 0))
                wd = self._lds_access(w, addr, 2, True, GROUPS_16, 32)
                for d in range(2):
                    self.lds[wd[act] + d] = rv(w, data[d])[act]
            w.stats["lds_ops"] -= 1
            w.lgkm.append(("ldsw", []))
        elif op == "ds_write2_b32":
            base = rv(w, A[0]).astype(np.int64)
            act = w.execmask()
            for which, data in ((0, A[1]), (1, A[2])):
                addr = base + 4 * int(M.get(f"offset{which}", 0))
                wd = self._lds_access(w, addr, 1, True, GROUPS_32, 32)
                self.lds[wd[act]] = rv(w, data)[act]
            w.stats["lds_ops"] -= 1
            w.lgkm.append(("ldsw", []))
        # ------------------------------------------------------------ VMEM (raw buffers, offen)
        elif op == "buffer_atomic_add":
            # 32-bit integer add in memory, no return value (sc0 clear); raw buffer, offen: same range check as a dword store
            data, voff, srd, soff = A
            base, nrec = self._srd(w, srd)
            so = rs(w, soff)
            assert M.get("offen") and not M.get("sc0")
            vo = rv(w, voff).astype(np.int64) + int(M.get("offset", 0))
            vals = rv(w, data)
            act = w.execmask()
            for l in range(LANES):
                if act[l] and vo[l] + so + 4 <= nrec and vo[l] >= 0:
                    addr = base + so + int(vo[l])
                    self.mem.write32(addr, (self.mem.read32(addr) + int(vals[l])) & 0xffffffff)
            w.vm.append([])
        elif op in ("buffer_load_dword", "buffer_load_dwordx2", "buffer_load_dwordx4", "buffer_store_dword", "buffer_store_dwordx2",
                    "buffer_store_dwordx4"):
            ndw = {"dword": 1, "dwordx2": 2, "dwordx4": 4}[op.split("_")[2]]
            if M.get("lds"):
                # LDS-DMA (`buffer_load_dword[x4] voffset, srsrc, soffset offen lds`): no data registers; lane l's 4 * ndw bytes land at
                # M0 + 4 * ndw * l (wave-uniform base, lane-linear image); M0 written by SALU needs one wait state first; the
                # bytes are in LDS once a counted vmcnt wait has retired the piece
                assert op.startswith("buffer_load") and ndw in (1, 4) and not M.get("offset")
                voff, srd, soff = A
                if self.check and w.state - w.m0_wr < 2:
                    raise SimError(f"wave {w.wid} pc {w.pc}: LDS-DMA right after M0 was written (needs s_nop 0)")
                base, nrec = self._srd(w, srd)
                so = rs(w, soff)
                assert M.get("offen")
                vo = rv(w, voff).astype(np.int64)
                act = w.execmask()
                lds_b = int(w.m0) + 4 * ndw * np.arange(LANES, dtype=np.int64)
                assert lds_b.max() + 4 * ndw <= 4 * len(self.lds) and int(w.m0) % 4 == 0
                dws, vals_all = [], []
                for d in range(ndw):
                    vals = np.zeros(LANES, dtype=U32)
                    ok_ = act & (vo + so + 4 * d + 4 <= nrec) & (vo + 4 * d >= 0)
                    if np.any(ok_):
                        vals[ok_] = self.mem.read32_vec(base + so + vo[ok_] + 4 * d)
                    dws.append((lds_b[act] // 4 + d))
                    vals_all.append(vals[act])
                w.vm.append(("dma", np.concatenate(dws), np.concatenate(vals_all)))
                w.stats["lds_dma"] = w.stats.get("lds_dma", 0) + 1
                w.state += cost
                w.pc = nxt
                return None
            data, voff, srd, soff = A
            for r_ in srd.regs() + (soff.regs() if isinstance(soff, Reg) else []):
                if self.check and r_[1] in w.valu_sgpr_wr and w.state - w.valu_sgpr_wr[r_[1]] < 5:
                    raise SimError(f"wave {w.wid} pc {w.pc}: VMEM reads s{r_[1]} {w.state - w.valu_sgpr_wr[r_[1]]} states after a VALU wrote it")
            base, nrec = self._srd(w, srd)
            so = rs(w, soff)
            if isinstance(voff, Sym):
                assert voff.name == "off" and not M.get("offen"), "off: no per-lane offset"
                vo = np.zeros(LANES, dtype=np.int64) + int(M.get("offset", 0))
            else:
                assert M.get("offen"), "offen addressing only"
                vo = rv(w, voff).astype(np.int64) + int(M.get("offset", 0))
            act = w.execmask()
            assert data.n == ndw
            if op.startswith("buffer_load"):
                self._chk_waw(w, data.regs())
                dst = w.v if data.kind == "v" else w.a
                for d in range(ndw):
                    vals = np.zeros(LANES, dtype=U32)
                    # gfx950 (scripts/probes/buffer_soffset.hip, profiles/r06/buffer_soffset_a.txt): the SCALAR offset is part of the
                    # range check of a raw buffer -- voffset + inst_offset + soffset + 4 <= num_records
                    ok_ = act & (vo + so + 4 * d + 4 <= nrec) & (vo + 4 * d >= 0)
                    if np.any(ok_):
                        vals[ok_] = self.mem.read32_vec(base + so + vo[ok_] + 4 * d)
                    dst[data.idx + d][act] = vals[act]
                    w.pending[(data.kind, data.idx + d)] = op
                w.vm.append(data.regs())
            else:
                for d in range(ndw):
                    vals = rv(w, data[d])
                    for l in range(LANES):
                        if act[l] and vo[l] + so + 4 * d + 4 <= nrec and vo[l] + 4 * d >= 0:
                            self.mem.write32(base + so + int(vo[l]) + 4 * d, int(vals[l]))
                w.vm.append([])
        else:
            raise SimError(f"unknown instruction {i.text()}")
        # VALU-result bookkeeping for the MFMA operand hazards
        if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_cmp") and op != "v_readfirstlane_b32":
            for x in (A[:2] if op == "v_swap_b32" else A[:1]):
                if isinstance(x, Reg):
                    for kk in x.regs():
                        w.valu_wr_state[kk] = w.state
        w.state += cost
        w.pc = nxt
        return None

    # ------------------------------------------------------------------ run to completion
    def run(self, order=None, max_steps=50_000_000):
        for w in self.waves:
            w.valu_wr_state = {}
        order = list(order or range(len(self.waves)))
        steps = 0
        while not all(w.done for w in self.waves):
            arrived = ended = 0
            for wi in order:
                w = self.waves[wi]
                if w.done:
                    continue
                while True:
                    r = self.step(w)
                    steps += 1
                    if steps > max_steps:
                        raise SimError("step limit")
                    if r == "barrier":
                        arrived += 1
                        break
                    if r == "end":
                        ended += 1
                        break
            if arrived and ended:
                raise SimError("some waves ended while others wait at a barrier")
            for w in self.waves:
                if not w.done:
                    w.epoch += 1
        return self
