"""Run a generated kernel through the interpreter on a small problem and compare with the accumulation-order model
(S_p = fmaf chain over each kc = 512 slice from +0, C = (..(0 + S_0) + S_1 ..) in order)."""
import struct
import sys
import time

import numpy as np

from . import f32_kernel as K
from .sim import Memory, Workgroup, fma32


def reference(A, B, kc, alpha=1.0, beta=0.0, C0=None):
    """C = (..((beta*C0) + alpha*S_0) + alpha*S_1 ..), every product and sum rounded to float32 (gemm_ukernel_generic.nim:53-76)"""
    M, Kd = A.shape
    N = B.shape[1]
    al = np.float32(alpha)
    run = np.zeros((M, N), dtype=np.float32) if beta == 0 else (np.float32(beta) * C0.astype(np.float32)).astype(np.float32)
    first = True
    for k0 in range(0, Kd, kc if kc else Kd):
        acc = np.zeros((M, N), dtype=np.float32)
        for k in range(k0, min(Kd, k0 + (kc if kc else Kd))):
            acc = fma32(A[:, k:k + 1] * np.ones((1, N), np.float32), np.ones((M, 1), np.float32) * B[k:k + 1, :], acc)
        run = (run + (al * acc).astype(np.float32)).astype(np.float32)
        first = False
    return run


def plan_units(tiles, Kd, G, exact, kc, BK, split=True):
    """host model of the launcher's unit arithmetic (gemm_f32_asm.cpp: plan_launch): P slices per tile, slice length, units per
    workgroup"""
    if not split:
        P, slen = 1, (Kd + BK - 1) // BK * BK
    elif exact:
        P, slen = (Kd + kc - 1) // kc, kc
    else:
        # one chain: slices of `split` K-tiles each (any multiple of BK is a legal cut)
        slen = BK * (split if isinstance(split, int) and split > 1 else 4)
        P = (Kd + slen - 1) // slen
    U = tiles * P
    assert 1 <= G <= U
    return P, slen, U // G, U % G


def sched_bytes(tm, tn, G, P=1, slen=1 << 30, q=None, r=0, noseed=0, ws=0, flags=0, group_m=None, xcd=False, two_level=False, strided=False, T=None):
    """the scheduler block of the kernel arguments (f32_kernel.py KA_SCHED); group_m None = one group: tile rows fastest;
    two_level: XCD x owns whole tiles, its G / 8 workgroups share them (launches that cut tiles); strided: workgroup v walks the whole
    tiles v, v + G, v + 2G ..."""
    gm = group_m or tm
    gsz_last = tm % gm or gm
    T = tm * tn if T is None else T       # (T: the tiles this launch covers -- the raster constants stay those of the whole grid)
    if strided:
        assert P == 1 or slen >= 1
        return struct.pack("<16IQQ", tm, tn, gm, gsz_last, K.magic_u32(gm * tn), K.magic_u32(gm), K.magic_u32(gsz_last), (G // 8) if xcd else 0,
                           (G % 8) if xcd else 0, P, K.magic_u32(P), G, T, slen, noseed | 4, K.magic_u32(G), ws, flags)
    if q is None:
        q, r = T * P // G, T * P % G
    if two_level:
        assert G % 8 == 0
        return struct.pack("<16IQQ", tm, tn, gm, gsz_last, K.magic_u32(gm * tn), K.magic_u32(gm), K.magic_u32(gsz_last), G // 8,
                           0, P, K.magic_u32(P), T, 0, slen, noseed | 2, K.magic_u32(G // 8), ws, flags)
    return struct.pack("<16IQQ", tm, tn, gm, gsz_last, K.magic_u32(gm * tn), K.magic_u32(gm), K.magic_u32(gsz_last), (G // 8) if xcd else 0,
                       (G % 8) if xcd else 0, P, K.magic_u32(P), q, r, slen, noseed, K.magic_u32(G), ws, flags)


def virtual_id(g, G, xcd):
    return (g % 8) * (G // 8) + min(g % 8, G % 8) + g // 8 if xcd and G >= 8 else g


BANK_MODEL = True      # LDS bank-conflict statistics (the CPU test suite switches them off: a quarter of the interpreter's time)


def run_grid(prog, mem, ka_, G, batch, lds_bytes, order=None, xcd=False):
    """every workgroup of a launch, one after the other: ascending virtual id, so that the running sum a workgroup receives (from the
    workgroup before it in unit order) is already in the workspace; returns the last workgroup's wave-0 statistics"""
    stats = None
    for bi in range(batch):
        for g in sorted(range(G), key=lambda g_: virtual_id(g_, G, xcd)):
            w = Workgroup(prog, mem, ka_, wg_id=(g, bi), lds_bytes=lds_bytes, bank_model=BANK_MODEL)
            w.run(order=order)
            stats = w.waves[0].stats
    return stats


def run_case(name, M, N, Kd, lda=None, ldb=None, ldc=None, seed=0, integer=False, order=None, over=None, verbose=True, alpha=1.0, beta=0.0,
             batch=1, bias=None, act=0, G=None, split=False, group_m=None, xcd=False, tol=None, noseed=0, two_level=False, csc=1, pre=0, interleaved=False,
             strided=False, mod=None, hybrid=False):
    """one f32 GEMM kernel through the interpreter; batch > 1: workgroup id y = batch index, operands `batch` spans apart.
    hybrid: the launcher's two-launch plan -- the whole rounds of the raster (tiles [0, T1), T1 = (T // G) * G) as a strided launch, the
    remaining tiles [T1, T) as a second launch of G2 = `hybrid` workgroups that cuts them along K (the kernels' tile base, KA_TAB).
    G: workgroups of the (persistent) launch, default one per tile; split: cut tiles along K at slice boundaries (laser-order: kc;
    one chain: `split` K-tiles per slice) so that the G workgroups get equal numbers of units"""
    g = (mod or K).make(name, **(over or {}))      # mod: another generator module with the f32 kernels' argument block (f32x16_kernel)
    g.build()
    c = g.c
    rng = np.random.default_rng(seed)
    lda, ldb, ldc = lda or Kd, ldb or N, ldc or N
    nt = c.b_kcontig
    if nt:
        ldb = ldb if (ldb and ldb >= Kd) else Kd
    LA = (M - 1) * lda + Kd
    LB = (N - 1) * ldb + Kd if nt else (Kd - 1) * ldb + N
    # (interleaved: rows are the fast direction -- a column-major-like view with a row stride; the kernels do NOT take it, the
    # launcher turns such a C into the transposed problem: kept here to show why, see the launcher's comment)
    if csc > 1 and not interleaved:
        ldc = max(ldc, (N - 1) * csc + 1)
    LC = (M - 1) * ldc + (N - 1) * csc + 1                  # C[i][j] at i * ldc + j * csc (a strided view, gemm_utils.nim:36-60)
    Aall, Ball, Call = np.zeros(batch * LA, np.float32), np.zeros(batch * LB, np.float32), np.full(batch * LC, np.nan, dtype=np.float32)
    As, Bs, C0s = [], [], []
    for b in range(batch):
        Af = np.zeros((M, lda), dtype=np.float32)
        Bf = np.zeros((N, ldb), dtype=np.float32) if nt else np.zeros((Kd, ldb), dtype=np.float32)
        Bm = rng.integers(-3, 4, (Kd, N)).astype(np.float32) if integer else rng.uniform(-0.1, 0.1, (Kd, N)).astype(np.float32)
        Af[:, :Kd] = rng.integers(-3, 4, (M, Kd)) if integer else rng.uniform(-0.1, 0.1, (M, Kd))
        Af[:, Kd:] = np.nan       # whatever lies between the rows must never reach the result
        if nt:
            Bf[:, :Kd] = Bm.T
            Bf[:, Kd:] = np.nan
        else:
            Bf[:, :N] = Bm
            Bf[:, N:] = np.nan
        # trim the allocations to exactly the operand spans: any read past them is a simulator error
        Aall[b * LA:(b + 1) * LA] = Af.reshape(-1)[:LA]
        Ball[b * LB:(b + 1) * LB] = Bf.reshape(-1)[:LB]
        C0 = None
        if beta != 0 and interleaved:
            C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
            Call[b * LC + (np.arange(M)[:, None] * ldc + np.arange(N)[None, :] * csc)] = C0
        elif beta != 0:           # C0 in the valid columns, NaN in the row padding
            C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
            full0 = np.full(M * ldc, np.nan, dtype=np.float32).reshape(M, ldc)
            full0[:, :(N - 1) * csc + 1:csc] = C0
            Call[b * LC:(b + 1) * LC] = full0.reshape(-1)[:LC]
        As.append(Af[:, :Kd].copy()); Bs.append(Bm); C0s.append(C0)
    tm, tn = (M + c.BM - 1) // c.BM, (N + c.BN - 1) // c.BN
    G = G or tm * tn
    P, slen, uq, ur = plan_units(tm * tn, Kd, min(G, tm * tn) if strided else G, c.exact, 512, c.BK, split)
    mem = Memory()
    a_, b_, c_ = mem.alloc(Aall), mem.alloc(Ball), mem.alloc(Call)
    ws_ = mem.alloc(np.full(G * g.tile_bytes() // 4, np.nan, dtype=np.float32))
    fl_base = mem.alloc(np.zeros(G + 1, dtype=np.uint32))      # [0] = the error word (receivers that gave up count themselves there)
    fl_ = fl_base + 4
    # fused epilogue: bias = "row" (1 x N, row stride 0), "col" (M x 1, column stride 0) or "full" (M x N, padded rows); act 1 = relu
    bias_ptr, rsb, csb, Bias = 0, 0, 0, None
    if bias:
        if bias == "row":
            Bias = rng.uniform(-1, 1, (1, N)).astype(np.float32); rsb, csb = 0, 1
        elif bias == "col":
            Bias = rng.uniform(-1, 1, (M, 1)).astype(np.float32); rsb, csb = 1, 0
        else:
            Bias = rng.uniform(-1, 1, (M, N)).astype(np.float32); rsb, csb = N, 1
        bias_ptr = mem.alloc(Bias.reshape(-1).copy())
    ka = struct.pack("<QQQQIIIIIIffQ", a_, b_, c_, 0, lda, ldb, ldc, M, N, Kd, alpha, beta, 0)
    ka += struct.pack("<Q", LA * 4) + b"\0" * 28 + struct.pack("<I", pre) + struct.pack("<QQ", LB * 4, LC * 4) + struct.pack("<QIIII", bias_ptr, rsb, csb, act, csc if csc != 1 else 0)
    t0 = time.time()
    if hybrid:
        T, G2 = tm * tn, int(hybrid)
        T1 = T // G * G
        assert batch == 1 and 0 < T1 < T and G2 >= 1
        ka1 = ka + sched_bytes(tm, tn, G, 1, (Kd + c.BK - 1) // c.BK * c.BK, None, 0, noseed, 0, 0, group_m, xcd, False, True, T=T1)
        ka1_ = mem.alloc(np.frombuffer(ka1, dtype=np.uint8))
        run_grid(g.p, mem, ka1_, G, 1, c.lds_alloc, order, xcd)
        R = T - T1
        P, slen, uq, ur = plan_units(R, Kd, G2, c.exact, 512, c.BK, split or True)
        ws2 = mem.alloc(np.full(G2 * g.tile_bytes() // 4, np.nan, dtype=np.float32))
        fl2 = mem.alloc(np.zeros(G2 + 1, dtype=np.uint32))
        kb = bytearray(ka)
        kb[24:28] = struct.pack("<I", T1)                      # KA_TAB's low word: the launch's first tile
        ka2 = bytes(kb) + sched_bytes(tm, tn, G2, P, slen, uq, ur, noseed, ws2, fl2 + 4, group_m, xcd or two_level, two_level, False, T=R)
        ka_ = mem.alloc(np.frombuffer(ka2, dtype=np.uint8))
        G, fl_base = G2, fl2
    else:
        ka += sched_bytes(tm, tn, G, P, slen, uq, ur, noseed, ws_, fl_, group_m, xcd or two_level, two_level, strided)
        assert len(ka) == K.KERNARG_SIZE
        ka_ = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    stats = run_grid(g.p, mem, ka_, G, batch, c.lds_alloc, order, xcd or two_level)
    flags_after = mem.get(fl_base, np.uint32, (G + 1,))
    if np.any(flags_after[1:]):
        raise AssertionError("a workspace flag was left set: the next launch would take a stale sum")
    run_case.last_error_word = int(flags_after[0])
    if flags_after[0] and not (noseed & 8):
        raise AssertionError("a receiver gave up waiting")
    got = mem.get(c_, np.float32, (batch * LC,))
    ok = pad_ok = True
    for b in range(batch):
        full = np.full(max(M * ldc, LC), np.nan, dtype=np.float32)
        full[:LC] = got[b * LC:(b + 1) * LC]
        if interleaved:
            idx = np.arange(M)[:, None] * ldc + np.arange(N)[None, :] * csc
            Cout = full[idx]
        else:
            Cout = full.reshape(M, ldc)[:, :(N - 1) * csc + 1:csc]
        relu = lambda x: np.where(x > 0, x, np.float32(0)).astype(np.float32)
        want = reference(relu(As[b]) if pre & 1 else As[b], relu(Bs[b]) if pre & 2 else Bs[b], 512 if c.exact else 0, alpha, beta, C0s[b])
        if Bias is not None:
            want = (want + np.broadcast_to(Bias, (M, N))).astype(np.float32)
        if act == 1:
            want = np.where(want > 0, want, np.float32(0)).astype(np.float32)
        if tol is None:
            ok &= bool(np.array_equal(Cout, want))
        else:       # (a cut one-chain launch adds partial sums: another rounding order than the single chain, by design)
            err = float(np.max(np.abs(Cout.astype(np.float64) - want.astype(np.float64))))
            ok &= 0.0 < err <= tol
        if interleaved:
            mask = np.ones(LC, dtype=bool)
            mask[idx.reshape(-1)] = False
            if beta == 0:
                pad_ok &= bool(np.all(np.isnan(full[:LC][mask])))
        elif ldc > N and csc == 1:
            pad_ok &= bool(np.all(np.isnan(full.reshape(M, ldc)[:, N:][:-1])))
        if csc > 1 and beta == 0 and not interleaved:          # the elements between the columns of the view are never written
            gaps = np.ones(ldc, dtype=bool)
            gaps[:(N - 1) * csc + 1:csc] = False
            pad_ok &= bool(np.all(np.isnan(full.reshape(M, ldc)[:-1][:, gaps])))
        if not ok and verbose:
            bad = np.argwhere(Cout != want)
            print("  batch", b, "first mismatches:", bad[:8].tolist(), Cout[tuple(bad[0])], want[tuple(bad[0])], "count", len(bad))
            break
    if verbose:
        print(f"{name} M={M} N={N} K={Kd} lda={lda} ldb={ldb} ldc={ldc} int={integer} batch={batch}: "
              f"{'OK' if ok and pad_ok else 'MISMATCH'}  ({time.time() - t0:.1f} s, bank-conflict cycles {stats['bank_conflict_cycles']}, "
              f"{stats['ins']} instr/wave, {stats['mfma']} mfma/wave)")
    return ok and pad_ok


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "exact_256x128x32"
    ok = run_case(name, 256, 128 if "128x32" in name else 256, 64, integer=True)
    sys.exit(0 if ok else 1)


def run_conv_case(name, images, Cin, H, W, M, pad, n_cut=None, seed=0, order=None, verbose=True, bias=False, act=0, kernel=(3, 3), stride=(1, 1), G=None):
    """implicit-GEMM convolution kernels (any kH x kW of up to Cfg.ntmax taps, any strides / zero padding / output width): every image
    through the interpreter, against im2col (conv2d_im2col.nim:62-87) + the slice-ordered model.  G: the unit-walking form of the
    kernel (Cfg.cpers), G workgroups over images x tiles units with pipelined transitions"""
    g = K.make(name, cpers=True) if G else K.make(name)
    g.build()
    c = g.c
    rng = np.random.default_rng(seed)
    pH, pW = (pad, pad) if isinstance(pad, int) else pad
    kH, kW = (kernel, kernel) if isinstance(kernel, int) else kernel
    sH, sW = (stride, stride) if isinstance(stride, int) else stride
    assert kH * kW <= c.ntmax
    oH, oW = (H + 2 * pH - kH) // sH + 1, (W + 2 * pW - kW) // sW + 1
    npix = oH * oW
    N = n_cut or npix
    Kd = Cin * kH * kW
    Kp = (Kd + 3) // 4 * 4          # (the launcher pads the filter rows to whole 16-byte pieces with zeros when Cin*kH*kW % 4 != 0)
    x = rng.uniform(-0.1, 0.1, (images, Cin, H, W)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (M, Kd)).astype(np.float32)
    wp = np.zeros((M, Kp), dtype=np.float32)
    wp[:, :Kd] = w
    out = np.full((images, M, npix), np.nan, dtype=np.float32)
    tm, tn = (M + c.BM - 1) // c.BM, (N + c.BN - 1) // c.BN
    mem = Memory()
    # the input is placed with nothing mapped directly before / after it: any access outside the tensor is an error
    a_, b_, c_ = mem.alloc(wp), mem.alloc(x), mem.alloc(out)
    Bias = rng.uniform(-1, 1, M).astype(np.float32) if bias else None
    bias_ptr = mem.alloc(Bias) if bias else 0
    NT = kH * kW
    ka = struct.pack("<QQQIIIIIIIIffQ", a_, b_, c_, K.magic_u32(NT), 0, Kp, 0, npix, M, N, Kp, 1.0, 0.0, 0)
    ka += struct.pack("<IIIIIIII", H, W, oW, pH, pW, Cin, npix, K.magic_u32(oW))
    ka += struct.pack("<IIQ", kH | kW << 8 | sH << 16 | sW << 24, NT | ((1024 + kW - 1) // kW) << 16, Cin * H * W * 4)
    ka += struct.pack("<Q", M * npix * 4)
    ka += struct.pack("<QIIII", bias_ptr, 1, 0, act, 0)
    if G:
        # units = images x tiles (an image's tiles stay together), workgroup g walks units g, g + G, ...
        T = tm * tn
        sb = bytearray(sched_bytes(tm, tn, T))
        sb[32:64] = struct.pack("<8I", 0, T, K.magic_u32(T), G, T * images, 0, 0, 0)
        ka += bytes(sb)
    else:
        ka += sched_bytes(tm, tn, tm * tn)        # one tile per workgroup, tile rows fastest (an image's pixels stay together)
    assert len(ka) == K.KERNARG_SIZE
    ka_ = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    t0 = time.time()
    if G:
        run_grid(g.p, mem, ka_, G, 1, c.lds_alloc, order)
    else:
        run_grid(g.p, mem, ka_, tm * tn, images, c.lds_alloc, order)
    got = mem.get(c_, np.float32, (images, M, npix))
    ok = True
    for img in range(images):
        xp = np.zeros((Cin, H + 2 * pH, W + 2 * pW), dtype=np.float32)
        xp[:, pH:pH + H, pW:pW + W] = x[img]
        Bm = np.stack([xp[ci, kh:kh + (oH - 1) * sH + 1:sH, kw:kw + (oW - 1) * sW + 1:sW].reshape(-1) for ci in range(Cin) for kh in range(kH) for kw in range(kW)])
        want = reference(w, Bm, 512 if c.exact else 0)
        if Bias is not None:
            want = (want + Bias[:, None]).astype(np.float32)
        if act == 1:
            want = np.where(want > 0, want, np.float32(0)).astype(np.float32)
        ok &= bool(np.array_equal(got[img][:, :N], want[:, :N]))
        ok &= bool(np.all(np.isnan(got[img][:, N:])))
        if not ok and verbose:
            bad = np.argwhere(got[img][:, :N] != want[:, :N])
            print("  image", img, "first mismatches", bad[:6].tolist(), "count", len(bad))
            break
    if verbose:
        print(f"{name} images={images} Cin={Cin} {H}x{W} M={M} k={kH}x{kW} s={sH}x{sW} pad={pad} N={N}/{npix}: {'OK' if ok else 'MISMATCH'} ({time.time() - t0:.1f} s)")
    return ok


def run_case64(name, M, N, Kd, lda=None, ldb=None, ldc=None, seed=0, order=None, over=None, verbose=True, alpha=1.0, beta=0.0, batch=1,
               G=None, split=False, group_m=None, xcd=False, noseed=0, two_level=False, strided=False):
    """float64 kernels (f64_kernel.py) through the interpreter.  Operands are small integers (alpha, beta dyadic), for which every
    product and partial sum is exact in float64 (the interpreter's f64 MFMA is mul + add, not an exact fma): this checks every
    address, layout, wait and hazard of the program; the accumulation ORDER with rounding is checked on hardware against the
    compiler-scheduled kernels and the CPU restatement.  batch > 1: workgroup id y = batch index, operands `batch` spans apart."""
    from . import f64_kernel as K64
    g = K64.make(name, **(over or {}))
    g.build()
    c = g.c
    nt = c.b_kcontig
    rng = np.random.default_rng(seed)
    lda, ldc = lda or Kd, ldc or N
    ldb = (ldb if (ldb and ldb >= Kd) else Kd) if nt else (ldb or N)
    LA, LC = (M - 1) * lda + Kd + 3, (M - 1) * ldc + N + 5          # spans between batches: a few elements of slack
    LB = ((N - 1) * ldb + Kd if nt else (Kd - 1) * ldb + N) + 7
    Aall, Ball, Call = np.full(batch * LA, np.nan), np.full(batch * LB, np.nan), np.full(batch * LC, np.nan)
    Am = rng.integers(-4, 5, (batch, M, Kd)).astype(np.float64)
    Bm = rng.integers(-4, 5, (batch, Kd, N)).astype(np.float64)
    C0 = rng.integers(-8, 9, (batch, M, N)).astype(np.float64)
    for b in range(batch):
        Af = np.full((M, lda), np.nan)
        Af[:, :Kd] = Am[b]
        if nt:
            Bf = np.full((N, ldb), np.nan)
            Bf[:, :Kd] = Bm[b].T
            Bflat = Bf.reshape(-1)[:(N - 1) * ldb + Kd]
        else:
            Bf = np.full((Kd, ldb), np.nan)
            Bf[:, :N] = Bm[b]
            Bflat = Bf.reshape(-1)[:(Kd - 1) * ldb + N]
        Aall[b * LA:b * LA + LA - 3] = Af.reshape(-1)[:(M - 1) * lda + Kd]
        Ball[b * LB:b * LB + LB - 7] = Bflat
        full0 = np.full((M, ldc), np.nan)
        if beta != 0:
            full0[:, :N] = C0[b]
        Call[b * LC:b * LC + LC - 5] = full0.reshape(-1)[:(M - 1) * ldc + N]
    tm, tn = (M + c.BM - 1) // c.BM, (N + c.BN - 1) // c.BN
    G = G or tm * tn
    if strided:
        G = min(G, tm * tn)         # workgroup v walks the whole tiles v, v + G, ... (pipelined transitions: f32_kernel.py Cfg.pipe)
    P, slen, uq, ur = plan_units(tm * tn, Kd, G, c.exact, 256, c.BK, split)
    mem = Memory()
    a_, b_, c_ = mem.alloc(Aall), mem.alloc(Ball), mem.alloc(Call)
    ws_ = mem.alloc(np.full(G * g.tile_bytes() // 8, np.nan))
    fl_base = mem.alloc(np.zeros(G + 1, dtype=np.uint32))      # [0] = the error word
    fl_ = fl_base + 4
    bs = (LA * 8, LB * 8, LC * 8) if batch > 1 else (0, 0, 0)
    ka = (struct.pack("<QQQQIIIIIIffQ", a_, b_, c_, 0, lda, ldb, ldc, M, N, Kd, 1.0, 0.0, 0) + struct.pack("<dd", alpha, beta)
          + struct.pack("<Q", bs[0]) + b"\0" * 16 + struct.pack("<QQ", bs[1], bs[2]) + b"\0" * 24)
    ka += sched_bytes(tm, tn, G, P, slen, uq, ur, noseed, ws_, fl_, group_m, xcd or two_level, two_level, strided)
    assert len(ka) == K.KERNARG_SIZE
    ka_ = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    t0 = time.time()
    stats = run_grid(g.p, mem, ka_, G, batch, c.lds_alloc, order, xcd or two_level)
    flags_after = mem.get(fl_base, np.uint32, (G + 1,))
    if np.any(flags_after[1:]):
        raise AssertionError("a workspace flag was left set")
    run_case64.last_error_word = int(flags_after[0])
    if flags_after[0] and not (noseed & 8):
        raise AssertionError("a receiver gave up waiting")
    got = mem.get(c_, np.float64, (batch * LC,))
    ok = True
    for b in range(batch):
        full = np.full(M * ldc, np.nan)
        full[:LC - 5] = got[b * LC:b * LC + LC - 5]
        full = full.reshape(M, ldc)
        want = alpha * (Am[b] @ Bm[b]) + (beta * C0[b] if beta != 0 else 0.0)
        okb = np.array_equal(full[:, :N], want) and (ldc == N or bool(np.all(np.isnan(full[:, N:][:-1])))) and bool(np.all(np.isnan(got[b * LC + LC - 5:(b + 1) * LC])))
        if not okb and verbose:
            bad = np.argwhere(full[:, :N] != want)
            print("  batch", b, "first mismatches:", bad[:8].tolist(), full[tuple(bad[0])] if len(bad) else None, want[tuple(bad[0])] if len(bad) else None, "count", len(bad))
        ok = ok and okb
    if verbose:
        print(f"f64 {name} M={M} N={N} K={Kd} lda={lda} ldb={ldb} ldc={ldc} alpha={alpha} beta={beta} batch={batch}: {'OK' if ok else 'MISMATCH'}  ({time.time() - t0:.1f} s, "
              f"bank-conflict cycles {stats['bank_conflict_cycles']}, {stats['ins']} instr/wave, {stats['mfma']} mfma/wave)")
    return ok


def pack_limb_tiles(X, np_=4):
    """host model of the packing pass (limb_planes.h, tile-major form): int32 / int64 matrix X[x][k] -> for every TR-row tile (128 for
    int32, 64 for int64) and 32-k tile a 16-KiB block [plane p][k half h][row r][16 bytes] of balanced base-256 digits; rows / k
    zero-padded to multiples of TR / 32"""
    tr = 128 if np_ == 4 else 64
    x, k = X.shape
    xp, kp = (x + tr - 1) // tr * tr, (k + 31) // 32 * 32
    P = np.zeros((xp, kp), dtype=np.uint64)
    mask = np.uint64((1 << (8 * np_)) - 1)
    P[:x, :k] = X.astype(np.int64).view(np.uint64) & mask
    bias = np.uint64(int("80" * (np_ - 1), 16))
    d = ((P + bias) & mask) ^ bias
    out = np.zeros((xp // tr, kp // 32, np_, 2, tr, 16), dtype=np.uint8)
    for pl in range(np_):
        digit = ((d >> np.uint64(8 * pl)) & np.uint64(0xff)).astype(np.uint8)          # [xp][kp]
        out[:, :, pl] = digit.reshape(xp // tr, tr, kp // 32, 2, 16).transpose(0, 2, 3, 1, 4)
    return out.reshape(-1), xp, kp


def run_case_i64(M, N, Kd, ldc=None, seed=0, order=None, verbose=True, alpha=1, beta=0, over=None):
    """int64 GEMM kernel (i8_kernel.py, "i64_64x64x32") through the interpreter: C = A B mod 2^64 with full-range int64 operands"""
    from . import i8_kernel as KI
    g = KI.make("i64_64x64x32", **(over or {}))
    g.build()
    c = g.c
    rng = np.random.default_rng(seed)
    A = rng.integers(-2**63, 2**63 - 1, (M, Kd), dtype=np.int64)
    B = rng.integers(-2**63, 2**63 - 1, (Kd, N), dtype=np.int64)
    ldc = ldc or N
    Ap, Mp, Kp = pack_limb_tiles(A, 8)
    Bp, Np_, _ = pack_limb_tiles(B.T.copy(), 8)
    KT = Kp // 32
    Cflat = np.full((M - 1) * ldc + N, 0x7bad7bad7bad7bad, dtype=np.uint64)
    C0 = rng.integers(0, 2**64, (M, N), dtype=np.uint64)
    if beta != 0:
        for r in range(M):
            Cflat[r * ldc:r * ldc + N] = C0[r]
    tm, tn = Mp // 64, Np_ // 64
    mem = Memory()
    a_, b_, c_ = mem.alloc(Ap), mem.alloc(Bp), mem.alloc(Cflat)
    ka = struct.pack("<QQQQIIIIIIiiQ", a_, b_, c_, 0, KT, 0, ldc, M, N, Kp, 1, 0, 0) + struct.pack("<qq", alpha, beta) + b"\0" * 64
    ka += sched_bytes(tm, tn, tm * tn, group_m=2, xcd=True)
    assert len(ka) == K.KERNARG_SIZE
    ka_ = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    t0 = time.time()
    stats = run_grid(g.p, mem, ka_, tm * tn, 1, c.lds_alloc, order, True)
    full = np.full(M * ldc, 0x7bad7bad7bad7bad, dtype=np.uint64)
    full[:len(Cflat)] = mem.get(c_, np.uint64, (len(Cflat),))
    full = full.reshape(M, ldc)
    Au, Bu = A.view(np.uint64), B.view(np.uint64)
    want = np.zeros((M, N), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k in range(Kd):
            want += Au[:, k:k + 1] * Bu[k:k + 1, :]          # uint64 arithmetic wraps mod 2^64
        want = want * np.uint64(alpha % 2**64)
        if beta != 0:
            want = want + C0 * np.uint64(beta % 2**64)
    ok = np.array_equal(full[:, :N], want) and (ldc == N or bool(np.all(full[:, N:][:-1] == 0x7bad7bad7bad7bad)))
    if verbose:
        print(f"i64 M={M} N={N} K={Kd} ldc={ldc}: {'OK' if ok else 'MISMATCH'}  ({time.time() - t0:.1f} s, bank-conflict cycles "
              f"{stats['bank_conflict_cycles']}, {stats['ins']} instr/wave, {stats['mfma']} mfma/wave)")
        if not ok:
            bad = np.argwhere(full[:, :N] != want)
            print("  first mismatches:", bad[:8].tolist(), hex(int(full[tuple(bad[0])])), hex(int(want[tuple(bad[0])])), "count", len(bad))
    return ok


def run_case_i32(M, N, Kd, ldc=None, seed=0, order=None, verbose=True, full_range=True, alpha=1, beta=0, over=None):
    """int32 GEMM kernel (i8_kernel.py) through the interpreter: C = A B mod 2^32 with full-range int32 operands"""
    from . import i8_kernel as KI
    g = KI.make(**(over or {}))
    g.build()
    c = g.c
    rng = np.random.default_rng(seed)
    lo, hi = (-2**31, 2**31) if full_range else (-100, 101)
    A = rng.integers(lo, hi, (M, Kd), dtype=np.int64).astype(np.int32)
    B = rng.integers(lo, hi, (Kd, N), dtype=np.int64).astype(np.int32)
    ldc = ldc or N
    Ap, Mp, Kp = pack_limb_tiles(A)
    Bp, Np_, _ = pack_limb_tiles(B.T.copy())
    KT = Kp // 32
    Cflat = np.full((M - 1) * ldc + N, 0x7bad7bad, dtype=np.uint32)
    C0 = rng.integers(0, 2**32, (M, N), dtype=np.uint64).astype(np.uint32)
    if beta != 0:
        full0 = np.full((M, ldc), 0x7bad7bad, dtype=np.uint32)
        full0[:, :N] = C0
        Cflat = full0.reshape(-1)[:(M - 1) * ldc + N].copy()
    tm, tn = Mp // 128, Np_ // 128
    mem = Memory()
    a_, b_, c_ = mem.alloc(Ap), mem.alloc(Bp), mem.alloc(Cflat)
    ka = struct.pack("<QQQQIIIIIIiiQ", a_, b_, c_, 0, KT, 0, ldc, M, N, Kp, alpha, beta, 0) + b"\0" * 80
    ka += sched_bytes(tm, tn, tm * tn, group_m=2, xcd=True)
    assert len(ka) == K.KERNARG_SIZE
    ka_ = mem.alloc(np.frombuffer(ka, dtype=np.uint8))
    t0 = time.time()
    stats = run_grid(g.p, mem, ka_, tm * tn, 1, c.lds_alloc, order, True)
    full = np.full(M * ldc, 0x7bad7bad, dtype=np.uint32)
    full[:len(Cflat)] = mem.get(c_, np.uint32, (len(Cflat),))
    full = full.reshape(M, ldc)
    want = ((A.astype(np.int64).astype(object) @ B.astype(np.int64).astype(object)) % (1 << 32)).astype(np.uint64).astype(np.uint32) \
        if M * N * Kd <= 200000 else None
    if want is None:
        acc = np.zeros((M, N), dtype=np.uint64)
        Au, Bu = A.astype(np.int64).astype(np.uint64) & np.uint64(0xffffffff), B.astype(np.int64).astype(np.uint64) & np.uint64(0xffffffff)
        for k in range(Kd):      # products mod 2^64 then mod 2^32: exact for the low 32 bits
            acc = (acc + (Au[:, k:k + 1] * Bu[k:k + 1, :])) & np.uint64(0xffffffff)
        want = acc.astype(np.uint32)
    want = ((want.astype(np.uint64) * np.uint64(alpha & 0xffffffff) + (C0.astype(np.uint64) * np.uint64(beta & 0xffffffff) if beta != 0 else np.uint64(0)))
            & np.uint64(0xffffffff)).astype(np.uint32)
    ok = np.array_equal(full[:, :N], want) and (ldc == N or bool(np.all(full[:, N:][:-1] == 0x7bad7bad)))
    if verbose:
        print(f"i32 M={M} N={N} K={Kd} ldc={ldc}: {'OK' if ok else 'MISMATCH'}  ({time.time() - t0:.1f} s, bank-conflict cycles "
              f"{stats['bank_conflict_cycles']}, {stats['ins']} instr/wave, {stats['mfma']} mfma/wave)")
        if not ok:
            bad = np.argwhere(full[:, :N] != want)
            print("  first mismatches:", bad[:8].tolist(), hex(int(full[tuple(bad[0])])), hex(int(want[tuple(bad[0])])), "count", len(bad))
    return ok
