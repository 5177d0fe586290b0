"""The hand-scheduled fp32 kernels (laser_amd/asmgen/f32_kernel.py) through the CPU interpreter (laser_amd/asmgen/sim.py):
every generated instruction stream is executed for small problems, wave by wave, and the result is compared BIT FOR BIT
with the accumulation-order model of SURVEY.md 8 (an fmaf chain per kc = 512 slice from +0, slices added in order --
gemm.nim:150-158 / gemm_ukernel_generic.nim:56-66).  The interpreter also enforces what the hardware does not forgive:
registers read while a load into them is still in flight, barriers passed with LDS writes un-waited, cross-wave LDS
races, the MFMA -> VALU and VALU-SGPR -> VMEM wait states, the M0 -> add-TID wait state, and every access outside an
operand's span (the operands are placed with nothing mapped around them; row padding is NaN).  No GPU needed."""
import pytest

from laser_amd.asmgen import check as C
from laser_amd.asmgen import f32_kernel as K

C.BANK_MODEL = False      # (bank-conflict statistics are printed by the command-line checker, nothing here asserts on them)


@pytest.mark.parametrize("name", sorted(K.CONFIGS))
def test_every_config_generates_and_carries_its_queue_state(name):
    g = K.make(name)
    g.build()                              # asserts the loop-carried VMEM / LDS queue state and the placement rules
    text = K.kernel_text(g, "lh_test")
    assert ".amdhsa_kernel lh_test" in text and "s_endpgm" in text
    assert g.p._v <= 256 and g.p._a <= 256 and g.p._s <= 100
    assert g.c.lds_alloc <= 160 * 1024


# (name, M, N, K, kwargs): one K-tile, several K-tiles, a kc fold (K > 512), K tails (K % BK != 0), ragged M / N, padded rows
GEMM_CASES = [
    ("exact_256x128x32", 70, 90, 64, {}),
    ("exact_256x128x32", 40, 36, 548, dict(lda=552, ldb=40, ldc=44)),           # fold + K tail + padded rows
    ("fast_256x256x16", 130, 60, 40, dict(ldb=64)),
    ("fast_256x128x32", 33, 130, 100, {}),
    ("exact_128x128x16", 129, 70, 36, {}),
    ("fast_128x128x16", 20, 140, 20, dict(ldc=144)),
    ("exact_256x128x32_nt", 50, 40, 68, dict(ldb=72)),
    ("fast_256x256x16_nt", 30, 270, 24, {}),
    ("exact_128x128x16_nt", 64, 64, 524, {}),                                   # fold, B transposed
    ("fast_128x128x16_nt", 140, 30, 12, {}),
    ("fast_256x128x32_nt", 60, 50, 36, {}),
]


# K not a multiple of 4: the piece that straddles K is zeroed element-wise in the staging registers (row padding is NaN here)
GEMM_CASES += [
    ("exact_256x128x32", 70, 90, 1030, dict(lda=1033, ldb=93, ldc=95)),
    ("fast_64x64x32_nt", 70, 90, 99, dict(lda=102, ldb=104)),
    ("exact_128x128x16_nt", 40, 50, 1031, dict(lda=1034, ldb=1036)),
    ("fast_256x256x16", 33, 270, 17, dict(lda=20)),
    ("exact_64x64x32", 70, 90, 6, dict(lda=9)),
]
# fused epilogue: C = act(beta * C0 + alpha * A B + bias): bias as a row / a column / a full view, relu
GEMM_CASES += [
    ("exact_256x128x32", 70, 90, 548, dict(bias="row", act=1, ldc=94)),
    ("fast_256x256x16", 70, 300, 40, dict(bias="col", act=1)),
    ("exact_64x64x32", 70, 90, 100, dict(bias="full", act=0, alpha=0.5, beta=2.0)),
    ("fast_128x128x16_nt", 140, 150, 36, dict(act=1)),
]
# C with a column stride (MatrixView, gemm_utils.nim:36-60): the epilogue's address arithmetic; nothing written between the columns
GEMM_CASES += [
    ("exact_64x64x32", 70, 90, 548, dict(csc=2, ldc=190)),
    ("exact_256x128x32", 70, 140, 100, dict(csc=3, alpha=0.5, beta=-1.0)),
    ("fast_256x256x16", 40, 300, 72, dict(csc=2, alpha=3.0, beta=0.5)),
    ("exact_64x64x32", 70, 90, 100, dict(csc=2, bias="row", act=1)),
]
# fused prologue (README.md:243-244): relu on A's / B's elements in the staging registers of the `_pre` variants (pre: bit 0 = A, 1 = B)
GEMM_CASES += [
    ("exact_64x64x32_pre", 70, 90, 548, dict(pre=1)),
    ("exact_64x64x32_pre", 70, 90, 548, dict(pre=2, alpha=0.5, beta=2.0)),
    ("exact_64x64x32_pre_nt", 70, 90, 550, dict(pre=3, lda=552, ldb=556)),
    ("exact_64x64x32_pre", 70, 90, 100, dict(pre=0)),
    ("fast_256x256x16_pre", 70, 300, 72, dict(pre=3)),
    ("fast_128x128x16_pre", 140, 130, 50, dict(pre=2, bias="row", act=1)),
]
# C = beta * C0 + alpha * A B: the running sum starts as beta * C0, every slice is scaled before it is added
GEMM_CASES += [
    ("exact_256x128x32", 70, 90, 1060, dict(alpha=0.75, beta=-1.5, ldc=100)),
    ("exact_64x64x32", 70, 40, 548, dict(alpha=-2.0, beta=1.0)),
    ("fast_256x256x16", 40, 300, 72, dict(alpha=3.0, beta=0.5)),
    ("fast_64x64x32_nt", 70, 90, 100, dict(alpha=0.5, beta=0.0)),
    ("exact_128x128x16_nt", 40, 50, 1028, dict(alpha=1.0, beta=0.25)),
]


@pytest.mark.parametrize("name,M,N,Kd,kw", GEMM_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}x{c[3]}" for c in GEMM_CASES])
def test_gemm_kernels_bit_exact_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case(name, M, N, Kd, verbose=False, **kw)


# Persistent launches (fewer workgroups than tiles) and tiles cut along K at slice boundaries (f32_kernel.py sched_next): G workgroups
# share tiles x P units.  A tile that straddles two workgroups is started by the one that owns its slice 0 -- first thing it does --
# which SENDS the running sum (beta * C0 included) through the workspace; the next workgroup does the tile's remaining slices last,
# on top of the RECEIVED sum: taken at the start of the piece if it has arrived (one run), else after the piece's first slice (two
# runs: noseed=1 forces that path).  Laser-order kernels must stay BIT-identical to the sequential slice order (gemm.nim:150-158)
# for any G; one-chain kernels are checked on integer-valued operands (their cut changes the rounding order, not the sum).
# xcd: the workgroup-id remap; group_m: grouped raster.
SPLIT_CASES = [
    ("exact_64x64x32", 70, 40, 1100, dict(G=4, split=True, alpha=-2.0, beta=1.0)),
    ("exact_64x64x32", 70, 90, 1100, dict(G=5, split=True)),
    ("exact_64x64x32", 70, 90, 1100, dict(G=2, split=True)),                       # whole tiles + one cut tile per workgroup
    ("exact_64x64x32", 70, 90, 1100, dict(G=3, split=False)),                      # persistent, whole tiles only
    ("exact_64x64x32_nt", 130, 70, 1030, dict(G=11, split=True, lda=1033, ldb=1036, ldc=75, alpha=0.75, beta=-1.5, group_m=2, xcd=True)),
    ("exact_64x64x32", 130, 70, 1030, dict(G=16, split=True, bias="row", act=1, xcd=True, group_m=2)),   # fused epilogue after the fix-up
    ("exact_128x128x16_nt", 140, 130, 519, dict(G=5, split=True, lda=522, ldb=524)),
    ("exact_256x128x32", 260, 130, 600, dict(G=3, split=True, alpha=0.75, beta=-1.5, ldc=150, group_m=1)),
    ("exact_64x64x32", 70, 40, 1100, dict(G=4, split=True, alpha=-2.0, beta=1.0, noseed=1)),
    ("exact_64x64x32", 70, 90, 1100, dict(G=2, split=True, noseed=1)),
    ("exact_64x64x32", 70, 40, 2100, dict(G=9, split=True, noseed=1, beta=0.5)),              # ranges inside one tile: receive, then send on
    ("exact_64x64x32", 70, 40, 2100, dict(G=9, split=True)),
    ("exact_64x64x32_nt", 130, 70, 1030, dict(G=11, split=True, lda=1033, ldb=1036, ldc=75, alpha=0.75, beta=-1.5, group_m=2, xcd=True, noseed=1)),
    ("exact_128x128x16", 140, 130, 600, dict(G=5, split=True, noseed=1, bias="col", act=1)),
    ("fast_64x64x32", 70, 40, 600, dict(G=9, split=2, integer=True, beta=2.0)),                # one chain, ranges inside one tile
    # two-level ranges (what the launcher uses for every cut launch): XCD x = id % 8 owns whole tiles, its G / 8 workgroups share them
    ("exact_64x64x32", 130, 200, 600, dict(G=16, split=True, two_level=True, group_m=2, beta=0.5)),
    ("exact_64x64x32", 130, 200, 600, dict(G=16, split=True, two_level=True, group_m=2, beta=0.5, noseed=1)),
    ("exact_64x64x32", 130, 70, 1100, dict(G=8, split=True, two_level=True, noseed=1)),
    ("fast_64x64x32", 130, 200, 300, dict(G=24, split=2, integer=True, two_level=True)),
    ("fast_64x64x32", 70, 90, 300, dict(G=5, split=2, alpha=0.5, beta=2.0, integer=True)),
    ("fast_64x64x32_nt", 70, 90, 301, dict(G=7, split=3, integer=True, lda=304, ldb=305, ldc=93)),
    ("fast_128x128x16", 140, 130, 200, dict(G=6, split=3, integer=True, beta=2.0, alpha=3.0)),
    ("fast_256x256x16", 270, 300, 100, dict(G=3, split=2, integer=True)),
]


@pytest.mark.parametrize("name,M,N,Kd,kw", SPLIT_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}x{c[3]}-G{c[4]['G']}{'-late' if c[4].get('noseed') else ''}{'-2lvl' if c[4].get('two_level') else ''}" for c in SPLIT_CASES])
def test_persistent_and_k_split_launches_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case(name, M, N, Kd, verbose=False, **kw)


# Pipelined tile transitions (Cfg.pipe, round 6): a persistent workgroup whose next run is another whole tile never leaves its K loop --
# the last two tile bodies of a tile fetch the next tile's first K-tiles, the next tile's first body (a transition body) finishes the
# old tile from the gaps behind its first MFMAs.  strided: workgroup v walks tiles v, v + G, v + 2G ... (what the launcher uses when
# there are more tiles than workgroup slots); the contiguous ranges of the cut plans pipeline their whole tiles the same way.  The
# result must be the same bits as one workgroup per tile: ragged M / N, strided C, alpha != 1, exactly 3 K-tiles (the switch happens in
# the transition body's own tail), folds inside a tile, several wave interleavings.
PIPE_CASES = [
    ("exact_64x64x32", 130, 130, 544, dict(G=3, strided=True)),
    ("exact_64x64x32", 130, 200, 96, dict(G=5, strided=True, ldc=205)),                       # 3 K-tiles: no body between two switches
    ("exact_64x64x32", 130, 200, 128, dict(G=4, strided=True, alpha=0.75, order=[3, 1, 0, 2])),
    ("exact_64x64x32_nt", 130, 130, 544, dict(G=2, strided=True, lda=548, ldb=552, xcd=True, group_m=2)),
    ("exact_256x128x32", 300, 260, 96, dict(G=2, strided=True, ldc=270)),
    ("exact_256x128x32", 300, 130, 544, dict(G=1, strided=True, alpha=-2.0)),                # one workgroup walks every tile; folds + transition
    ("exact_256x128x32_nt", 300, 260, 160, dict(G=3, strided=True, csc=2)),
    ("exact_128x128x16", 140, 260, 560, dict(G=2, strided=True, csc=2)),
    ("exact_128x128x32", 260, 390, 160, dict(G=4, strided=True, order=[2, 0, 3, 1])),
    ("fast_256x128x32", 300, 260, 128, dict(G=2, strided=True, alpha=0.5)),
    ("fast_256x256x16", 300, 520, 64, dict(G=2, strided=True)),
    ("fast_128x128x16_nt", 140, 390, 80, dict(G=3, strided=True, ldc=400)),
    ("fast_64x64x32", 130, 200, 160, dict(G=7, strided=True, xcd=True, group_m=2)),
    # contiguous ranges (the cut plans): whole tiles between the head and the tail piece of a range are pipelined too
    ("exact_64x64x32", 130, 130, 544, dict(G=5, split=True)),
    ("exact_64x64x32", 130, 130, 544, dict(G=8, split=True, two_level=True, noseed=1)),
    # launches that may NOT pipeline take the ordinary path: beta != 0, a bias, a K tail
    ("exact_64x64x32", 130, 130, 544, dict(G=3, strided=True, beta=0.5)),
    ("exact_64x64x32", 130, 130, 544, dict(G=3, strided=True, bias="row", act=1)),
    ("exact_64x64x32", 130, 130, 556, dict(G=3, strided=True)),
    ("fast_256x128x32", 300, 260, 64, dict(G=2, strided=True)),                               # two K-tiles only
]


@pytest.mark.parametrize("name,M,N,Kd,kw", PIPE_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}x{c[3]}-G{c[4]['G']}-{i}" for i, c in enumerate(PIPE_CASES)])
def test_pipelined_tile_transitions_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case(name, M, N, Kd, verbose=False, **kw)


def test_pipelined_transitions_are_really_taken():
    """the transition bodies run (tiles - workgroups) times per wave, the ordinary epilogue once per workgroup"""
    from laser_amd.asmgen import sim
    seen = {"trans": 0, "done": 0}
    orig = sim.Workgroup.step

    def step(self, w):
        i = self.ins[w.pc]
        if i.op == "label" and w.wid == 0:
            if "trans_s" in i.args[0]:
                seen["trans"] += 1
            elif i.args[0].startswith(".L_done_"):
                seen["done"] += 1
        return orig(self, w)
    sim.Workgroup.step = step
    try:
        assert C.run_case("exact_64x64x32", 130, 200, 576, verbose=False, G=3, strided=True)      # 3 x 4 tiles on 3 workgroups
    finally:
        sim.Workgroup.step = orig
    assert seen == {"trans": 9, "done": 3}


DEEP_CASES = [("exact_64x64x32", 70, 50, 33, {}), ("exact_64x64x32", 64, 64, 96, {}), ("exact_64x64x32", 70, 50, 97, dict(alpha=0.5, beta=0.25)),
              ("exact_64x64x32", 70, 50, 131, {}), ("exact_64x64x32", 64, 64, 1057, {}), ("fast_64x64x32_nt", 70, 90, 161, dict(alpha=0.5)),
              ("exact_64x64x32_nt", 130, 70, 1100, dict(G=5, split=True)), ("fast_64x64x32", 100, 100, 545, dict(G=3, split=5, integer=True))]


@pytest.mark.parametrize("name,M,N,Kd,kw", DEEP_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}x{c[3]}" for c in DEEP_CASES])
def test_two_tile_prefetch_option_in_the_interpreter(name, M, N, Kd, kw):
    """Cfg.deep (two sets of staging registers, a K-tile requested two bodies ahead; no shipped kernel uses it -- it measured no
    faster): one, two, three, four and many K-tiles, ragged K, K % 4 != 0, folds, a cut launch"""
    assert C.run_case(name, M, N, Kd, verbose=False, over={"deep": True}, **kw)


def test_one_chain_split_matches_within_rounding():
    """random data through a cut one-chain launch: not bit-equal to the single chain by design (the order did change), equal within
    float32 rounding (the sum did not)"""
    assert C.run_case("fast_64x64x32", 70, 90, 300, G=5, split=2, verbose=False, tol=1e-5)


def test_interpreter_reports_a_receive_that_precedes_its_send():
    """the interpreter runs workgroups one after the other: a workgroup simulated BEFORE the one that sends it its running sum must be
    reported (spin on a flag nobody has set), not hang -- this is also what a lost flag store would look like"""
    from laser_amd.asmgen import check
    from laser_amd.asmgen.sim import SimError
    real = check.virtual_id
    try:
        check.virtual_id = lambda g, G, xcd: -g          # descending order: receivers first
        with pytest.raises(SimError):
            check.run_case("exact_64x64x32", 70, 40, 1100, G=4, split=True, verbose=False)
    finally:
        check.virtual_id = real


F64_SPLIT_CASES = [
    ("exact_64x64x16", 70, 80, 530, dict(lda=532, G=7, split=True, alpha=2.0, beta=0.5, xcd=True, group_m=1)),
    ("fast_64x64x16_nt", 70, 80, 150, dict(G=6, split=2, alpha=2.0, beta=0.5)),
    ("fast_128x128x16", 130, 70, 86, dict(G=3, split=2)),
    ("exact_128x128x16", 130, 140, 300, dict(G=5, split=True, alpha=-1.0, beta=2.0)),
    ("exact_64x64x16", 70, 80, 530, dict(lda=532, G=7, split=True, alpha=2.0, beta=0.5, noseed=1)),
    ("exact_64x64x16_nt", 70, 40, 1040, dict(G=9, split=True, noseed=1, beta=-1.0)),
]


@pytest.mark.parametrize("name,M,N,Kd,kw", F64_SPLIT_CASES, ids=[f"f64-{c[0]}-{c[1]}x{c[2]}x{c[3]}-G{c[4]['G']}{'-late' if c[4].get('noseed') else ''}" for c in F64_SPLIT_CASES])
def test_f64_persistent_and_k_split_launches_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case64(name, M, N, Kd, verbose=False, **kw)


# (name, images, Cin, H, W, M, pad, n_cut): padding 1 / 0 / asymmetric / 2, the kc fold, ragged last pixel tile, pixel cut
CONV_CASES = [
    ("conv_exact_256x128x32", 2, 8, 12, 16, 40, 1, None),
    ("conv_fast_256x128x32", 1, 4, 10, 12, 24, 0, None),
    ("conv_exact_256x128x32", 1, 64, 6, 8, 20, 1, None),                     # K = 576: a fold and a K tail
    ("conv_fast_256x128x32", 1, 20, 14, 20, 16, (0, 1), 128),                # main launch of a main + tail split
    ("conv_exact_256x128x32", 1, 12, 9, 18, 260, (2, 2), None),              # two row tiles, two pixel tiles (ragged)
]


@pytest.mark.parametrize("name,images,Cin,H,W,M,pad,n_cut", CONV_CASES,
                         ids=[f"{c[0]}-{c[1]}x{c[2]}x{c[3]}x{c[4]}-pad{c[6]}" for c in CONV_CASES])
def test_conv_kernels_bit_exact_in_the_interpreter(name, images, Cin, H, W, M, pad, n_cut):
    assert C.run_conv_case(name, images, Cin, H, W, M, pad, n_cut=n_cut, verbose=False)


# Round 6: kernel size, strides, padding and output width are RUN-TIME values of the convolution kernels (conv2d_im2col.nim:42-88 is
# generic in all of them): the tap table is built by a scalar loop, the tap arithmetic of the loop is scalar.  (name, images, Cin, H, W,
# M, pad, n_cut, kernel, stride): 5x5; the 7x7 stride-2 first layer (C_in = 3: K = 147, filter rows zero-padded to whole 16-byte
# pieces); 3x3 stride 2 with odd output widths (a lane's pixel pair straddles two output rows); 1x1; non-square and even kernels;
# 42 and 49 taps (the smaller tiles' table); the reference's stride-2 KAT geometry (conv2d_common.nim:188-283).
CONV_GEOMETRY_CASES = [
    ("conv_exact_256x128x32", 1, 8, 11, 13, 40, 2, None, 5, 1),
    ("conv_fast_64x128x32", 1, 3, 20, 22, 20, 3, None, 7, 2),
    ("conv_exact_128x128x32", 2, 24, 9, 9, 70, 1, None, 3, 2),
    ("conv_fast_128x128x32", 1, 40, 6, 7, 30, 0, None, 1, 1),
    ("conv_exact_256x128x32", 1, 6, 9, 15, 30, (0, 3), None, (1, 7), 1),
    ("conv_exact_64x128x32", 1, 6, 15, 9, 30, (3, 0), None, (7, 1), (2, 1)),
    ("conv_exact_256x128x32", 1, 70, 8, 8, 20, 1, None, (2, 4), (1, 2)),        # K = 560: a fold
    ("conv_fast_128x128x32", 1, 2, 30, 31, 12, 2, None, (6, 7), 3),
    ("conv_fast_128x128x32", 1, 2, 40, 45, 12, 2, 128, 7, (2, 3)),              # 49 taps, the main part of a cut launch
    ("conv_exact_256x128x32", 1, 3, 5, 5, 2, 1, None, 3, 2),
    ("conv_exact_64x128x32", 1, 16, 43, 4, 40, (2, 0), None, (1, 3), (2, 3)),   # output width 1 (found by scripts/fuzz_conv.py: pix / oW with oW == 1)
]


@pytest.mark.parametrize("name,images,Cin,H,W,M,pad,n_cut,kernel,stride", CONV_GEOMETRY_CASES,
                         ids=[f"{c[0]}-{c[2]}x{c[3]}x{c[4]}-k{c[8]}-s{c[9]}" for c in CONV_GEOMETRY_CASES])
def test_conv_kernels_any_geometry_in_the_interpreter(name, images, Cin, H, W, M, pad, n_cut, kernel, stride):
    assert C.run_conv_case(name, images, Cin, H, W, M, pad, n_cut=n_cut, kernel=kernel, stride=stride, verbose=False)


# Round 6: the convolution kernels as unit walkers (Cfg.cpers; `_p` kernels): G workgroups over images x tiles units, the transition
# from one unit to the next inside the K loop when K is a multiple of 32 with three K-tiles or more and the epilogue is the plain store
# (the next unit's tap table, gathers and filter pieces requested before the old tile's last two K-tiles are multiplied; its C stores from
# the gaps of the new tile's first K-tile body), through the ordinary epilogue otherwise.
CONV_WALK_CASES = [
    # (name, images, Cin, H, W, M, pad, kernel, stride, G, bias)
    ("conv_fast_64x128x32_p", 2, 32, 17, 18, 40, 1, 3, 1, 2, False),              # 6 units (ragged last tile of each image) on 2 workgroups
    ("conv_exact_64x128x32_p", 1, 64, 17, 18, 70, 1, 3, 1, 2, False),             # K = 576: a fold inside every tile + transitions, two row tiles
    ("conv_exact_256x128x32_p", 3, 32, 12, 12, 200, 1, 3, 1, 4, False),           # 6 units on 4 workgroups (2 + 2 + 1 + 1)
    ("conv_fast_256x128x32_p", 3, 32, 12, 12, 260, 0, (1, 3), (2, 1), 2, False),
    ("conv_exact_128x128x32_p", 2, 96, 20, 13, 130, 0, 1, 1, 3, False),           # 1x1: K = 96, exactly the three K-tiles the switch needs
    ("conv_fast_128x128x32_p", 3, 12, 17, 18, 40, 1, 3, 1, 4, False),             # K = 108: not a multiple of 32 -> every unit through the epilogue
    ("conv_exact_64x128x32_p", 2, 32, 17, 18, 40, 1, 3, 1, 2, True),              # bias + relu: not pipelined either
    ("conv_exact_128x128x32_p", 1, 64, 9, 9, 130, 1, 3, 1, 1, False),             # one workgroup, both units of the image
]


@pytest.mark.parametrize("name,images,Cin,H,W,M,pad,kernel,stride,G,bias", CONV_WALK_CASES)
def test_conv_unit_walkers_bit_exact_in_the_interpreter(name, images, Cin, H, W, M, pad, kernel, stride, G, bias):
    assert C.run_conv_case(name, images, Cin, H, W, M, pad, kernel=kernel, stride=stride, G=G, bias=bias, act=1 if bias else 0, verbose=False)


def test_conv_unit_walkers_take_the_pipelined_transition(monkeypatch):
    """the transition body is really what stores a walked tile: without its stores the result is wrong (and with K not a multiple
    of the K-tile the same kernel never gets there)"""
    from laser_amd.asmgen import f32_kernel as K
    monkeypatch.setattr(K.Gen, "trans_after", lambda self, b: None)
    assert not C.run_conv_case("conv_fast_64x128x32_p", 2, 32, 17, 18, 40, 1, G=2, verbose=False)
    assert C.run_conv_case("conv_fast_64x128x32_p", 2, 12, 17, 18, 40, 1, G=2, verbose=False)


def test_conv_kernels_fused_bias_relu_in_the_interpreter():
    assert C.run_conv_case("conv_exact_256x128x32", 2, 8, 12, 16, 40, 1, bias=True, act=1, verbose=False)
    assert C.run_conv_case("conv_fast_64x128x32", 1, 64, 6, 8, 70, (0, 1), bias=True, act=0, verbose=False)


# Round 6: the launcher's hybrid plan -- whole rounds of the raster as a strided launch, the remaining tiles as a second launch that cuts
# them along K; the kernels add the launch's first tile (KA_TAB's low word) to the scheduler's relative tile numbers
HYBRID_CASES = [
    ("exact_64x64x32", 130, 200, 600, dict(G=5, hybrid=3)),                                        # 12 tiles: 10 strided + 2 tiles x 2 slices on 3 workgroups
    ("fast_64x64x32", 130, 200, 160, dict(G=5, hybrid=4, tol=1e-3)),
    ("exact_128x128x16", 260, 260, 560, dict(G=4, hybrid=2)),                                      # 9 tiles: 8 + 1 tile whose two slices go to two workgroups
    ("exact_64x64x32", 130, 380, 600, dict(G=10, hybrid=8, two_level=True, group_m=2)),            # 18 tiles: 10 + 8 tiles on 8 workgroups, one per XCD
    ("exact_64x64x32_nt", 130, 264, 600, dict(G=7, hybrid=2, group_m=2, noseed=1)),                # 15 tiles: 14 + 1; the late receive path
]


@pytest.mark.parametrize("name,M,N,Kd,kw", HYBRID_CASES, ids=[f"hybrid-{c[0]}-{c[1]}x{c[2]}x{c[3]}" for c in HYBRID_CASES])
def test_hybrid_two_launch_plan_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case(name, M, N, Kd, verbose=False, **kw)


def test_hybrid_plan_16x16_block_tile_in_the_interpreter():
    from laser_amd.asmgen import f32x16_kernel as K16
    assert C.run_case("exact_96x96x32", 200, 300, 544, G=5, hybrid=3, mod=K16, verbose=False)      # 12 tiles: 10 + 2


# float64 kernels (f64_kernel.py): integer-valued operands (exact in f64: the interpreter's f64 MFMA is mul + add)
F64_CASES = [
    ("fast_64x64x16", 70, 90, 48, {}),
    ("exact_64x64x16", 40, 30, 530, dict(lda=532)),                   # kc = 256 folds + K tail
    ("fast_128x128x16", 33, 50, 22, dict(lda=24, ldb=52, ldc=54)),
    ("exact_128x128x16", 130, 70, 290, {}),
    ("exact_64x64x16", 70, 80, 300, dict(lda=310, ldc=90, batch=3, alpha=2.0, beta=0.5)),   # grid y = batch index
    ("fast_64x64x16_nt", 64, 70, 32, dict(batch=2)),
]


@pytest.mark.parametrize("name,M,N,Kd,kw", F64_CASES, ids=[f"f64-{c[0]}-{c[1]}x{c[2]}x{c[3]}" for c in F64_CASES])
def test_f64_kernels_exact_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case64(name, M, N, Kd, verbose=False, **kw)


# Round 6: pipelined tile transitions in the float64 kernels (f64_kernel.py trans_after / pipe_c_addr; the strided plan): workgroup v
# walks the whole tiles v, v + G, ...; K a multiple of 16 with three K-tiles or more and beta == 0 go from tile to tile inside the K
# loop, anything else through the ordinary epilogue of the same kernel
F64_PIPE_CASES = [
    ("fast_64x64x16", 130, 130, 64, dict(G=3)),
    ("exact_64x64x16", 130, 130, 288, dict(G=3)),                       # a kc fold inside every tile + transitions
    ("exact_64x64x16_nt", 130, 130, 288, dict(G=5, xcd=True)),
    ("exact_128x128x16", 260, 130, 272, dict(G=1)),                     # one workgroup walks all three tiles (the last one ragged)
    ("fast_128x128x16_nt", 140, 390, 48, dict(G=2)),                    # exactly three K-tiles
    ("exact_64x64x16", 130, 130, 290, dict(G=3)),                       # K not a multiple of 16: not pipelined
    ("exact_64x64x16", 130, 130, 288, dict(G=3, alpha=0.5)),
    ("exact_64x64x16", 130, 130, 288, dict(G=3, beta=2.0)),             # beta != 0: not pipelined
]


@pytest.mark.parametrize("name,M,N,Kd,kw", F64_PIPE_CASES, ids=[f"f64-pipe-{c[0]}-{c[1]}x{c[2]}x{c[3]}-{i}" for i, c in enumerate(F64_PIPE_CASES)])
def test_f64_pipelined_tile_transitions_in_the_interpreter(name, M, N, Kd, kw):
    assert C.run_case64(name, M, N, Kd, verbose=False, strided=True, **kw)


def test_f64_pipelined_transitions_are_really_taken(monkeypatch):
    from laser_amd.asmgen import f64_kernel as K64
    monkeypatch.setattr(K64.Gen64, "trans_after", lambda self, b: None)
    assert not C.run_case64("fast_64x64x16", 130, 130, 64, G=3, strided=True, verbose=False)


def test_f64_configs_generate():
    from laser_amd.asmgen import f64_kernel as K64
    for name in K64.CONFIGS:
        g = K64.make(name)
        g.build()
        assert g.p._v <= 256 and g.p._a <= 256 and g.c.lds_alloc <= 160 * 1024
        assert "v_mfma_f64_16x16x4_f64" in K64.kernel_text(g, "lh_test64")


@pytest.mark.parametrize("M,N,Kd,ldc", [(128, 128, 224, None), (70, 200, 100, 204), (130, 129, 33, None)])
def test_i32_limb_kernel_exact_in_the_interpreter(M, N, Kd, ldc):
    """int32 GEMM mod 2^32 on int8 digit planes (i8_kernel.py), full-range operands, against exact integer arithmetic; the host
    model of the packing pass (check.pack_limb_tiles) is what limb_planes.h implements on the device"""
    assert C.run_case_i32(M, N, Kd, ldc=ldc, verbose=False)


@pytest.mark.parametrize("M,N,Kd,ldc", [(64, 64, 64, None), (70, 90, 100, 93)])
def test_i64_limb_kernel_exact_in_the_interpreter(M, N, Kd, ldc):
    """int64 GEMM mod 2^64 on eight int8 digit planes (36 limb products, 64-bit recombination with carries), full-range operands"""
    assert C.run_case_i64(M, N, Kd, ldc=ldc, verbose=False)


@pytest.mark.parametrize("which,M,N,Kd,ldc", [("i32", 130, 129, 100, 133), ("i64", 70, 90, 100, 93)])
def test_integer_limb_kernels_register_staged_loop_still_exact(which, M, N, Kd, ldc):
    """the shipped integer kernels bring their operand blocks into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, what the cases above
    run); the register-staged loop of rounds 4-5 stays as a generator option (A/B timing, ablations) and must keep the same bits"""
    run = C.run_case_i32 if which == "i32" else C.run_case_i64
    assert run(M, N, Kd, ldc=ldc, verbose=False, over=dict(dma=False))


def test_interpreter_holds_lds_dma_bytes_back_until_the_counted_wait():
    """an LDS-DMA piece is in LDS only once a vmcnt wait has retired it (a ds_read issued earlier sees the OLD bytes, no stall): with the
    counted waits dropped the fragments are read from stages that have not landed -- the result must be wrong, not silently right"""
    assert not C.run_case_i32(128, 128, 224, verbose=False, over=dict(ablate=("vmwaits",)))


@pytest.mark.parametrize("alpha,beta", [(-3, 0), (0x123456789abcdef, -0x7654321fedcba987), (1, 5)])
def test_i64_limb_kernel_alpha_beta_in_the_interpreter(alpha, beta):
    """C = alpha * A B + beta * C0 mod 2^64 in the int64 kernel's epilogue (three bodies: plain, alpha only, alpha and beta)"""
    assert C.run_case_i64(70, 50, 40, ldc=61, alpha=alpha, beta=beta, seed=4, verbose=False)


def test_interpreter_rejects_a_read_of_a_register_still_loading():
    """the checks are live: dropping the counted waits must be caught, not silently pass"""
    from laser_amd.asmgen.sim import SimError
    with pytest.raises((SimError, AssertionError)):
        C.run_case("exact_256x128x32", 40, 40, 96, verbose=False, over=dict(ablate=("vmwaits",)))


def test_receivers_that_give_up_count_themselves_in_the_error_word():
    """flags bit 3 (tests): every receiver of a cut launch takes the path of a hand-over that timed out -- one atomic add per workgroup
    to the error word in front of the flags -- and goes on; the launcher reads the word back behind the launch and fails the next
    call on the stream (tests/test_gpu_scheduler.py).  In the interpreter the senders have run (workgroups execute in unit order),
    so the results are still right: what is checked here is the count, the flags left clear, and that C is stored (srdC intact)."""
    assert C.run_case("exact_64x64x32", 70, 90, 1100, G=5, split=True, noseed=8, verbose=False) and C.run_case.last_error_word == 3
    assert C.run_case("exact_64x64x32", 70, 90, 1100, G=5, split=True, noseed=9, verbose=False) and C.run_case.last_error_word == 3
    assert C.run_case("fast_64x64x32", 70, 40, 600, G=9, split=2, integer=True, beta=2.0, noseed=8, verbose=False) and C.run_case.last_error_word == 8
    assert C.run_case64("exact_64x64x16", 70, 40, 600, G=5, split=True, noseed=8, verbose=False) and C.run_case64.last_error_word == 3
    assert C.run_case("exact_64x64x32", 70, 90, 1100, G=5, split=True, verbose=False) and C.run_case.last_error_word == 0


# Round 6: the 16x16-block tile family (laser_amd/asmgen/f32x16_kernel.py: v_mfma_f32_16x16x4_f32, tiles 96x96 and 160x96 -- the
# reference's bench shape 1920^3 is 240 tiles of 160x96 on 256 CUs, gemm_bench_float32.nim:383-410).  Same checks as the 32x32-block
# kernels: bit-exact against the slice-ordered fmaf model, folds, K tails (K % 32 != 0), ragged M / N, padded rows (NaN between them),
# alpha / beta, B transposed, batches, persistent launches with K-slice hand-overs.
X16_CASES = [
    ("fast_96x96x32", 100, 110, 64, {}),
    ("exact_96x96x32", 100, 110, 548, dict(lda=552, ldb=116, ldc=120)),
    ("exact_96x96x32", 70, 90, 1060, dict(alpha=0.75, beta=-1.5, ldc=100)),
    ("fast_96x96x32", 33, 130, 100, dict(alpha=3.0, beta=0.5)),
    ("exact_96x96x32_nt", 97, 100, 524, dict(lda=528, ldb=532)),
    ("fast_96x96x32_nt", 140, 30, 12, {}),
    ("exact_160x96x32", 170, 100, 548, dict(ldc=104)),
    ("fast_160x96x32", 161, 97, 36, dict(beta=2.0)),
    ("exact_160x96x32_nt", 170, 100, 580, dict(alpha=-2.0, beta=1.0)),
    ("fast_160x96x32_nt", 40, 50, 96, {}),
    ("exact_96x96x32", 4, 4, 4, {}),
    ("exact_96x96x32", 100, 110, 64, dict(batch=2)),
    ("exact_96x96x32", 100, 100, 620, dict(G=2, split=True)),
    ("exact_96x96x32", 100, 200, 548, dict(G=3, split=True, alpha=0.75, beta=-1.5, noseed=1)),
    ("exact_96x96x32_nt", 200, 200, 548, dict(G=8, split=True, two_level=True, group_m=2)),
    ("fast_96x96x32", 100, 200, 300, dict(G=5, split=2, integer=True, beta=2.0)),
    ("exact_160x96x32", 170, 100, 600, dict(G=3, split=True, group_m=1)),
    ("exact_96x96x32", 200, 200, 96, dict(G=2, strided=True)),
    ("exact_128x96x32", 130, 100, 548, dict(lda=552, ldb=104, ldc=108)),
    ("fast_128x96x32_nt", 129, 97, 100, dict(alpha=3.0, beta=0.5)),
    ("exact_192x96x32_nt", 200, 100, 548, dict(G=2, split=True)),
    ("fast_192x96x32", 193, 97, 36, {}),
    ("exact_160x160x32", 170, 100, 548, dict(ldc=104)),
    ("exact_160x160x32_nt", 170, 170, 548, dict(G=2, split=True, alpha=0.75, beta=-1.5)),
    # pipelined tile transitions on this family: strided whole-tile plans, folds inside a tile, one chain, exactly three
    # K-tiles, launches that may not pipeline (beta != 0, a K tail)
    ("exact_96x96x32", 200, 300, 96, dict(G=2, strided=True, ldc=304)),
    ("exact_96x96x32", 200, 100, 576, dict(G=2, strided=True, alpha=0.75)),
    ("fast_96x96x32_nt", 200, 300, 128, dict(G=4, strided=True)),
    ("exact_160x160x32_nt", 330, 170, 96, dict(G=1, strided=True)),
    ("exact_192x96x32", 390, 200, 96, dict(G=3, strided=True, xcd=True, group_m=2)),
    ("exact_96x96x32", 100, 200, 576, dict(G=1, strided=True, beta=0.5)),
    ("exact_96x96x32", 100, 200, 580, dict(G=1, strided=True)),
    # fused epilogue C = act(beta * C0 + alpha * A B + bias): bias as a row / a column / a full view, relu; five block columns (160x160)
    ("exact_96x96x32", 100, 110, 548, dict(bias="row", act=1, ldc=114)),
    ("fast_96x96x32", 100, 210, 40, dict(bias="col", act=1)),
    ("exact_160x96x32", 170, 100, 100, dict(bias="full", act=0, alpha=0.5, beta=2.0)),
    ("fast_160x160x32_nt", 170, 165, 36, dict(act=1)),
    ("exact_96x96x32", 200, 300, 96, dict(G=3, strided=True, bias="row", act=1)),
    # C with a column stride (MatrixView, gemm_utils.nim:36-60): nothing written between the columns
    ("exact_96x96x32", 70, 90, 548, dict(csc=2, ldc=190)),
    ("fast_160x160x32", 40, 300, 72, dict(csc=2, alpha=3.0, beta=0.5)),
    ("exact_192x96x32_nt", 300, 260, 160, dict(G=3, strided=True, csc=2)),
]


@pytest.mark.parametrize("name,M,N,Kd,kw", X16_CASES, ids=[f"x16-{c[0]}-{c[1]}x{c[2]}x{c[3]}-{i}" for i, c in enumerate(X16_CASES)])
def test_16x16_block_kernels_bit_exact_in_the_interpreter(name, M, N, Kd, kw):
    from laser_amd.asmgen import f32x16_kernel as K16
    assert C.run_case(name, M, N, Kd, verbose=False, mod=K16, **kw)


def test_16x16_block_kernels_generate_within_the_register_and_lds_budget():
    from laser_amd.asmgen import f32x16_kernel as K16
    for name in K16.CONFIGS:
        g = K16.make(name)
        g.build()
        assert g.p._v <= 256 and g.p._a <= 256 and g.p._s <= 100 and g.c.lds_alloc <= 160 * 1024, name
        assert "v_mfma_f32_16x16x4_f32" in K16.kernel_text(g, "lh_test")
