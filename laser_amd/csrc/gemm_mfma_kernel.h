// laser_amd/csrc/gemm_mfma_kernel.h -- fp32 / fp64 GEMM on the gfx950 exact-IEEE matrix cores
// (v_mfma_f32_32x32x2_f32, v_mfma_f64_16x16x4_f64: both are bitwise k-ordered fma chains).
//
// GPU re-mapping of Laser's Goto/BLIS nest (gemm.nim:109-176), not a translation of it:
//
//   Laser (CPU)                                   here (MI355X)
//   ------------------------------------------    --------------------------------------------------
//   ic loop over mc=192 row blocks, omp for        one workgroup per BM x BN tile of C, XCD-aware
//   jr loop over NR panels, omp taskloop           grouped raster so neighbours share L2 panels
//   pack_A_mc_kc / pack_B_kc_nc into L2/L3 panel   global->LDS staging IS the packing: both operand
//   buffers, strides resolved while packing        tiles land in LDS as k-major panels T[k][x]
//   (gemm_packing.nim:24-94)                       (= Laser's A~[k][ii] / B~[k][jj] layout) whatever
//                                                  the source strides; ragged edges zero-filled like
//                                                  the reference's zero-padded panels
//   pc loop over kc=512 slices, C += per slice     K loop inside the workgroup, 3-stage LDS ring;
//   (gemm.nim:150-158)                             in LASER_ORDER mode the MFMA accumulator restarts
//                                                  every kc and slices are folded into a running C
//                                                  in ascending order -> bit-identical to Laser
//   MR x NR register micro-kernel, k-ascending     (BM/WM) x (BN/WN) wave tile of 32x32
//   FMA (gemm_ukernel_generator.nim:140-250)       v_mfma_f32_32x32x2_f32 blocks; an f32 MFMA is
//                                                  bitwise a k-ordered fmaf chain
//   scalar epilogue, beta==0 never reads C         fused epilogue on the accumulator registers,
//   (gemm_ukernel_generic.nim:53-126)              same beta==0 / beta==1 / alpha==1 case split
//
// LDS panel image: T[k][x ^ swz(k)], swz(k) = ((k>>2)&7) << SWZ_SHIFT.  Chosen so that ALL of
//   (a) fragment reads  (32 lanes = 32 consecutive x at one k, ds_read_b32)        are conflict-free,
//   (b) 16-B vector writes along x (ds_write_b128, unit-stride-in-x operands)      are conflict-free,
//   (c) transposing scalar writes (4 consecutive k of one x per lane, ds_write_b32) are conflict-free
// with no padding (see DESIGN.md section 3 for the bank arithmetic).
// fp32 operands of the 3-stage configurations use the k-quad image instead when both can (kq_swz / kq_row below):
// [x][BK] rows whose 16-byte chunks hold what one MFMA half consumes over four k-steps -> ds_read_b128 fragments.
//
// Operand loaders (TileLoader): plain 16-byte vector loads; EDGE = the same through a bounds-checked buffer
// descriptor (any alignment, ragged extents); GEN = scalar, any strides; LOAD_IM2COL / LOAD_CONV_PATCH = the
// implicit-GEMM convolution's B operand (per-element gather / LDS-resident input patch).
#pragma once
#ifndef LH_KQ
#define LH_KQ 1  // k-quad LDS image for k-contiguous fp32 operands (0: the k-major image everywhere, for A/B runs)
#endif
#ifndef LH_FOLD_SCALAR
#define LH_FOLD_SCALAR 1  // alpha == 1 laser-order slice fold as scalar v_add_f32 (0: vector-typed -> v_pk_add_f32, for A/B runs)
#endif
#include <type_traits>

#include "common.h"

// Ablation switches of the main loop (timing only, results are wrong when one is on): 1 = no HBM loads of later tiles,
// 2 = no LDS stores, 4 = no mid-tile barrier.  Production builds: the DBG template flag is false and the test is compiled
// away.  -DLH_DBG_MASK=<bits> builds an experimental library whose PRODUCTION instantiations have the pieces removed at
// compile time (scripts/ablate_exact.py) -- unlike the run-time g.dbg switches of the probe instantiation, that leaves the
// steady-state loop one basic block with its counted waits intact, so the deltas are the pieces' real prices.
#ifdef LH_DBG_MASK
#define LH_DBG_KEEP(bit) (!((LH_DBG_MASK) & (bit)))
#else
#define LH_DBG_KEEP(bit) (!DBG || !(g.dbg & (bit)))
#endif

namespace laser_hip {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;
using f64x2 = __attribute__((ext_vector_type(2))) double;

// Per-element-type description of the matrix instruction and of a 16-byte memory piece.
//   EPV   elements per 16-byte vector           MB   edge of one MFMA output block
//   KS    k consumed by one MFMA                ACC  accumulator elements per lane per block
//   lane -> operand element: A[x = lx(lane)][k = lk(lane)], B[k = lk(lane)][x = lx(lane)]
//   accumulator element r of lane -> C[acc_row(r, lane)][acc_col(lane)]
template <typename E>
struct Mma;
template <>
struct Mma<float> {
  using Vec = f32x4;
  using Acc = f32x16;
  static constexpr int EPV = 4, MB = 32, KS = 2, ACC = 16, WGRP = 32;  // WGRP: lanes per ds_write bank group
  static constexpr int KC = 512;                                       // gemm_tiling.nim:310: 2048 / sizeof(float32)
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int lx(int lane) { return lane & 31; }
  static __device__ __forceinline__ int lk(int lane) { return lane >> 5; }
  static __device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 31; }
  // (hipcc contracts a*b+c across __fmul_rn/__fadd_rn under its default -ffp-contract=fast; the pragma drops
  // the `contract` flag from these instructions, which survives inlining)
  static __device__ __forceinline__ float mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
  }
  static __device__ __forceinline__ float add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
  }
  // run + alpha*ab on a whole accumulator block, unfused (two roundings, like the scalar epilogue);
  // vector-typed so that it lowers to v_pk_mul_f32 / v_pk_add_f32 (8 + 8 instructions per block)
  static __device__ __forceinline__ float tanh_(float x) { return tanhf(x); }
  static __device__ __forceinline__ float exp_(float x) { return expf(x); }
  static __device__ __forceinline__ Acc fold_acc(Acc run, Acc ab, float alpha) {
#pragma clang fp contract(off)
    const Acc t = ab * alpha;
    return run + t;
  }
  // alpha == 1: 1*x is x bit for bit, the multiply is dropped; 16 scalar v_add_f32 per accumulator block, as
  // inline asm so that the SLP vectoriser cannot pair them up: packed f32 VALU beside MFMAs costs ~13 cycles per
  // instruction more than the scalar form (MI355X_MICROARCH.md "price of one filler beside MFMAs"), and this
  // block rides between two MFMAs of the fold tile
  static __device__ __forceinline__ Acc fold_acc1(Acc run, Acc ab) {
#if LH_FOLD_SCALAR
#pragma unroll
    for (int e = 0; e < ACC; e++) {
      float r = run[e];
      asm("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(ab[e]));
      run[e] = r;
    }
    return run;
#else
#pragma clang fp contract(off)
    return run + ab;
#endif
  }
};
template <>
struct Mma<double> {
  using Vec = f64x2;
  using Acc = f64x4;
  static constexpr int EPV = 2, MB = 16, KS = 4, ACC = 4, WGRP = 16;
  static constexpr int KC = 256;                                       // 2048 / sizeof(float64)
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int lx(int lane) { return lane & 15; }
  static __device__ __forceinline__ int lk(int lane) { return lane >> 4; }
  static __device__ __forceinline__ int acc_row(int r, int lane) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
  static __device__ __forceinline__ double mul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
  }
  static __device__ __forceinline__ double add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
  }
  static __device__ __forceinline__ double tanh_(double x) { return tanh(x); }
  static __device__ __forceinline__ double exp_(double x) { return exp(x); }
  static __device__ __forceinline__ Acc fold_acc(Acc run, Acc ab, double alpha) {
#pragma clang fp contract(off)
    const Acc t = ab * alpha;
    return run + t;
  }
  static __device__ __forceinline__ Acc fold_acc1(Acc run, Acc ab) {
#pragma clang fp contract(off)
    return run + ab;
  }
};

// LDS panel swizzle: element (k, x) lives at T[k][x ^ swz(k)], swz(k) = ((k / EPV) & 7) << SHIFT.
// A transposing write group (WGRP lanes = BK/EPV k-pieces x WGRP/(BK/EPV) consecutive x) then covers
// WGRP distinct banks; SHIFT >= log2(EPV) keeps a 16-byte piece along x contiguous; swz < MB keeps a
// fragment inside its MFMA block.
template <typename E, int BK>
struct SwzShift {
  static constexpr int NKQ = BK / Mma<E>::EPV;
  static constexpr int value = (Mma<E>::WGRP / NKQ >= 16) ? 4 : (Mma<E>::WGRP / NKQ >= 8) ? 3 : (Mma<E>::WGRP / NKQ >= 4) ? 2
                               : (Mma<E>::WGRP / NKQ >= 2) ? 1 : 0;
  static_assert(NKQ >= 1 && NKQ <= 8 && ((NKQ - 1) << value) < Mma<E>::MB, "unsupported BK for this element type");
};

template <typename E, int BK>
__device__ __forceinline__ int swz(int k) {
  return ((k / Mma<E>::EPV) & 7) << SwzShift<E, BK>::value;
}

// ---- operand tile loader: HBM -> registers -> LDS panel (the "packing" stage) -------------------
// Every HBM load is UNCONDITIONAL and comes from a clamped, always-valid address; what must read as
// zero (k beyond K -- Laser zero-pads its panels the same way, gemm_packing.nim:46-55,85-94 -- and
// the convolution's padding) is recorded in a per-piece validity mask and zeroed when the piece is
// written to LDS one K-tile later.  A select (or a predicated load) at LOAD time would make the wave
// wait for the HBM round trip in the middle of the MFMA stream (measured: 70 vs 120 TFLOP/s on the
// ragged conv-shaped GEMM).  Rows/cols beyond M/N need no zeroing: they only feed outputs that are
// never stored.
// k-quad LDS image (KQ_, fp32 operands with unit stride along k): row-major [x][BK] where every group of 8 k
// is stored as (k0 k2 k4 k6 | k1 k3 k5 k7), i.e. two 16-byte chunks holding what MFMA half hi = 0 / 1 of a
// lane consumes over FOUR consecutive k-steps -- one ds_read_b128 per fragment per 4 k-steps instead of four
// ds_read_b32 (256 vs 128 B/clk and a quarter of the instructions, MI355X_MICROARCH.md LDS table), and the
// 16-byte global piece (4 consecutive k) lands with two ds_write_b64 instead of four transposing
// ds_write_b32.  Chunk index is XOR-swizzled with kq_swz(x) so the 16 lanes of every ds_read_b128 lane group
// ({0-3,12-15,20-27}, ...: all 16 residues of x mod 16) cover the 64 banks exactly once.
// (the XOR with bit log2(R/2) of x, R = 64/BK rows per 64-bank line, keeps the read property -- lanes of equal
// x mod R still see distinct values -- and sends rows x and x + R/2, which share a 16-lane ds_write_b64 group
// when k-contiguous pieces are stored, to opposite chunk parities: SQ_LDS_BANK_CONFLICT 256 -> 0 cycles per tile)
template <int BK>
__device__ __forceinline__ int kq_swz(int x) {
  constexpr int R = 64 / BK;
  return ((x / R) ^ ((x / (R / 2)) & 1)) % (BK / 4);
}
// physical row of x: neighbouring rows swap inside every second x quad, so that the pair-mode store of an
// x-contiguous operand (all 16 lanes of a ds_write_b64 group write rows of ONE parity) still reaches both
// halves of the 32 write banks.  scripts/kq_bank_check.py replays the three access patterns against the bank
// rules of MI355X_MICROARCH.md: 0 conflicts for BK = 16 and 32 (without the swap: 4 cycles per pair store at BK = 16,
// which SQ_LDS_BANK_CONFLICT confirmed: 128 cycles per 256x256x16 tile).
__device__ __forceinline__ int kq_row(int x) { return x ^ ((x >> 2) & 1); }

template <typename E, int BX, int BK, int NT, int MODE, bool KQ_ = false>
struct TileLoader {
  using M_ = Mma<E>;
  using Vec = typename M_::Vec;
  static constexpr int EPV = M_::EPV;
  static constexpr int NV = (BX * BK / EPV) / NT;  // 16-B pieces per thread per tile
  static_assert(NV >= 1 && (BX * BK / EPV) % NT == 0, "tile must split evenly over the workgroup");
  static constexpr bool ALONG_K = (MODE == LOAD_VEC_K || MODE == LOAD_GEN_K || MODE == LOAD_VEC_K_EDGE);
  static constexpr bool KQ = KQ_;
  static_assert(!KQ || (std::is_same<E, float>::value && (BK == 16 || BK == 32) && BX % 16 == 0 &&
                        (ALONG_K || MODE == LOAD_VEC_X || MODE == LOAD_VEC_X_EDGE || MODE == LOAD_IM2COL)),
                "k-quad image: fp32, 16-byte pieces along k or (transposed on the way into LDS) along x");
  // x-contiguous operand into the k-quad image: pieces are handled in PAIRS (rows k and k + 2 of the same x
  // quad, held by one thread), because k and k + 2 are neighbours in a chunk -> element e of both pieces is
  // one ds_write_b64 into row x = 4xq + e (4 writes per pair, where single pieces needed 8 ds_write_b32).
  static constexpr bool PAIRS = KQ && !ALONG_K;
  static constexpr int GROUPN = PAIRS ? 2 : 1;  // pieces per op group
  static_assert(!PAIRS || NV % 2 == 0, "pair mode needs an even number of pieces per thread");
  // piece i -> (x quad, k) for pieces along x.  Plain: 32 consecutive lanes walk x (512 B of one k row).
  // Pair mode: lane bits (a:2 = x quad, p:1, h:1, rest) -> k = 8J + 4h + p + 2*(i & 1); the 16 contiguous lanes
  // of a ds_write_b64 group then hit 4 swizzle classes x 2 chunk parities x 2 word pairs = 16 banks x 2 (free).
  static __device__ __forceinline__ void piece_xk(int t, int i, int &xq, int &k) {
    if constexpr (PAIRS) {
      const int idx = t + (i / 2) * NT;
      const int a = idx % 4, p = (idx / 4) % 2, h = (idx / 8) % 2, c = idx / 16;
      xq = (c % (BX / 16)) * 4 + a;
      k = 8 * (c / (BX / 16)) + 4 * h + p + 2 * (i & 1);
    } else {
      const int idx = t + i * NT;
      xq = idx % (BX / EPV);
      k = idx / (BX / EPV);
    }
  }
  static constexpr bool EDGE = (MODE == LOAD_VEC_X_EDGE || MODE == LOAD_VEC_K_EDGE);
  static constexpr bool VEC = (MODE == LOAD_VEC_X || MODE == LOAD_VEC_K || EDGE);
  static constexpr bool CONV = (MODE == LOAD_IM2COL);
  static constexpr bool PATCH = (MODE == LOAD_CONV_PATCH);
  static constexpr bool MASKED = !(MODE == LOAD_VEC_X || MODE == LOAD_VEC_K || CONV || PATCH);  // CONV / PATCH zero through the buffer's bounds check
  static_assert(!(CONV || PATCH) || std::is_same<E, float>::value, "the implicit-GEMM loaders are fp32 only");
  static_assert(ALONG_K || SwzShift<E, BK>::value >= (EPV == 4 ? 2 : 1), "16-byte pieces along x must stay contiguous");
  Vec v[NV];
  uint32_t msk[MASKED ? NV : 1];  // bit c: element c of piece i is real data (else it reads as zero)
  // LOAD_IM2COL state.  The output pixel a lane gathers for is fixed for the whole K loop, so everything
  // that depends on it is decoded ONCE (init_conv): per piece element the linear offset of its window
  // origin  org = (oh*sH - pH)*W + (ow*sW - pW)  (may be negative) and two bitmaps, bit kr of `rowbad` /
  // bit kq of `colbad` = "input row oh*sH-pH+kr / column ow*sW-pW+kq does NOT exist" (padding; all ones for
  // x beyond N) -- packed 4 x (8 + 8) bits per piece; kernels up to 8x8, larger ones take the
  // explicit-workspace path.
  // The k side (channel c, kernel row kr, kernel column kq of the k this piece gathers) is a running
  // state advanced by BK per tile with carries: no division in the K loop.  load_op() must therefore be
  // called for consecutive tiles k0 = 0, BK, 2*BK, ... exactly once each (it is).
  // The image is read through a raw buffer descriptor: an element that must read as zero gets the offset
  // 0xfffffffc (all ones before the *4), which the bounds check answers with 0 -- as it does for k beyond K
  // (channel >= C lies past num_records).  No clamping, no validity mask, no select at LDS-store time.
  // (probe: scripts/probes/buffer_oob.hip -- per-dword check at the top end; a negative offset zeroes the
  // WHOLE multi-dword load, which is why every element carries its own offset.)
  int32_t org[CONV ? NV : 1][CONV ? 4 : 1];
  uint32_t bad[CONV ? NV : 1][CONV ? 2 : 1];  // [i][e/2], 16 bits per element: rowbad | colbad << 8
  int32_t kc_[CONV ? NV : 1], kr_[CONV ? NV : 1], kq_[CONV ? NV : 1];
  __amdgpu_buffer_rsrc_t rsrc;  // this image as a bounds-checked buffer (wave-uniform, lives in SGPRs)

  // EDGE modes read the operand panel through a raw buffer descriptor: 16-byte loads need only element
  // alignment (the probe scripts/probes/buffer_oob.hip: dword-aligned b128 loads work, each dword is range
  // checked on its own at the top end), and whatever lies beyond the operand's last element reads as 0
  // instead of faulting -- so ragged extents, odd leading dimensions and unaligned bases all take the vector
  // path, with no address clamping.  `span`: elements from `panel` to the operand's last valid element + 1.
  __device__ __forceinline__ void init_buf(const E *panel, int64_t span) {
    if constexpr (EDGE) {
      const uint64_t b = reinterpret_cast<uint64_t>(panel);
      const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
      const int64_t bytes = span * (int64_t)sizeof(E);
      const int nrec = __builtin_amdgcn_readfirstlane((int)(uint32_t)(bytes > 0xffffffffll ? 0xffffffffll : (bytes < 0 ? 0 : bytes)));
      rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bu), 0, nrec, 0x00020000);
    }
  }
  __device__ __forceinline__ Vec buf_load(uint32_t byte_off) const {
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return __builtin_bit_cast(Vec, q);
  }

  // LOAD_CONV_PATCH: the B "tile" of the implicit-GEMM convolution is an input PATCH -- for each of the cCHM
  // channels a K-tile can touch, the cRp input rows the tile's output pixels reach, stored as rows of cPWs >= W + 8 (conv_patch_geom, common.h)
  // floats (input column j at index 4 + j; everything else, the zero padding, stays 0 from the one-time clear).
  // The patch rows are contiguous in HBM: each thread owns up to NV 16-byte pieces (channel slot, row, column
  // quad), decoded once; per K-tile only the first channel c0 moves (running state, no division).  Rows outside
  // the image and channels beyond C get offset 0xfffffffc: the buffer's bounds check returns zeros.
  // The kH*kW shifted views of the patch are produced by the fragment reads (kernel, ldgroup), not here.
  int32_t pg_off[PATCH ? NV : 1];   // element offset of the piece inside channel 0 of the image, or -1: never valid
  int32_t pg_ch[PATCH ? NV : 1];    // channel slot of the piece
  int32_t pl_dst[PATCH ? NV : 1];   // LDS destination (floats from the patch base), 16-byte aligned
  int32_t p_c0, p_rem;              // first channel of the tile being loaded, and (tile's first k) mod kH*kW
  // offset table: for every k of a tile, where its (channel slot, kernel row, kernel column) view starts in the
  // patch -- written next to the patch by the store pass (thread t owns k = t % BK, a running state like the
  // gather loader's), read by the fragment gathers: no index arithmetic in the MFMA loop.
  int32_t tw_c0, tw_rem, tk_c, tk_r, tk_q;
  static constexpr int TBL = BX * BK - BK;  // table = the last BK words of the B region of a stage (4 dummy words before it)
  __device__ __forceinline__ void init_patch(const GemmArgs<E> &g, int64_t n0, int t, const E *image, int kbase = 0) {
    if constexpr (PATCH) {
      const uint64_t b = reinterpret_cast<uint64_t>(image);
      const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
      const int khw = g.ckH * g.ckW, C = (int)(g.K / khw);
      const int bytes = __builtin_amdgcn_readfirstlane(C * g.cH * g.cW * 4);
      rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bu), 0, bytes, 0x00020000);
      const int row0 = (int)(n0 / g.coW) * g.csH - g.cpH;  // first input row of the patch
      const int w4 = g.cW / 4, per_ch = g.cRp * w4, total = g.cCHM * per_ch;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int p = t + i * NT;
        const int ch = p / per_ch, rr = (p - ch * per_ch) / w4, j = p - ch * per_ch - rr * w4;
        const int row = row0 + rr;
        const bool ok = p < total && (unsigned)row < (unsigned)g.cH;
        pg_off[i] = ok ? row * g.cW + 4 * j : -1;
        pg_ch[i] = ch;
        pl_dst[i] = p < total ? ch * (g.cRp * g.cPWs) + rr * g.cPWs + 4 + 4 * j : TBL - 4;  // spare pieces: a dummy slot (no predicated store)
      }
      // (kbase: first k this workgroup computes -- 0, or the start of its kc slice in the K-slice-parallel form)
      p_c0 = kbase / khw;
      p_rem = kbase - p_c0 * khw;
      tw_c0 = p_c0;
      tw_rem = p_rem;
      const int kl = t % BK + kbase;
      tk_c = kl / khw;
      tk_r = (kl - tk_c * khw) / g.ckW;
      tk_q = kl - tk_c * khw - tk_r * g.ckW;
    }
  }
  // store pass of one tile (called once per tile, tiles in order): this thread's k offset into the patch
  __device__ __forceinline__ void store_table(E *__restrict__ lds, int t, const GemmArgs<E> *cg) {
    if constexpr (PATCH) {
      const int khw = cg->ckH * cg->ckW;
      const int dc = min(max(tk_c - tw_c0, 0), cg->cCHM - 1);  // k beyond K stays inside the patch (A is zero there)
      const int koff = dc * (cg->cRp * cg->cPWs) + tk_r * cg->cPWs + tk_q;
      const int kl = t % BK;
      reinterpret_cast<int32_t *>(lds)[TBL + (kl & 1) * (BK / 2) + (kl >> 1)] = koff;  // [hi][k-step]; same value from every owner of kl
      int nq = tk_q + cg->cdq, nr = tk_r + cg->cdr, nc = tk_c + cg->cdc;
      if (nq >= cg->ckW) { nq -= cg->ckW; nr++; }
      if (nr >= cg->ckH) { nr -= cg->ckH; nc++; }
      tk_q = nq; tk_r = nr; tk_c = nc;
      const int r = tw_rem + cg->cdr * cg->ckW + cg->cdq;
      const int w = (r >= khw) ? 1 : 0;
      tw_rem = r - w * khw;
      tw_c0 = tw_c0 + cg->cdc + w;
    }
  }

  __device__ __forceinline__ void init_conv(const GemmArgs<E> &g, int64_t n0, int t, const E *image, int kbase = 0) {
    if constexpr (CONV) {
      // readfirstlane: tell the compiler the descriptor is uniform (else every load becomes a waterfall loop)
      const uint64_t b = reinterpret_cast<uint64_t>(image);
      const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);  // (int results: no sign extension)
      const int khw = g.ckH * g.ckW;
      const int bytes = __builtin_amdgcn_readfirstlane((int)(g.K / khw) * g.cH * g.cW * 4);
      rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bu), 0, bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < NV; i++) {
        int xq, k;
        piece_xk(t, i, xq, k);
        k += kbase;
        kc_[i] = k / khw;
        const int rem = k - kc_[i] * khw;
        kr_[i] = rem / g.ckW;
        kq_[i] = rem - kr_[i] * g.ckW;
        bad[i][0] = bad[i][1] = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int64_t j = n0 + 4 * xq + e;
          uint32_t m = 0xffffu;
          org[i][e] = 0;
          if (j < g.N) {
            const int oh = (int)(j / g.coW), ow = (int)(j - (int64_t)oh * g.coW);
            const int r0 = oh * g.csH - g.cpH, c0 = ow * g.csW - g.cpW;
            org[i][e] = r0 * g.cW + c0;
            m = 0;
            for (int q = 0; q < g.ckH; q++) m |= ((unsigned)(r0 + q) < (unsigned)g.cH ? 0u : 1u) << q;
            for (int q = 0; q < g.ckW; q++) m |= ((unsigned)(c0 + q) < (unsigned)g.cW ? 0u : 1u) << (8 + q);
          }
          bad[i][e >> 1] |= m << (16 * (e & 1));
        }
      }
    }
  }

  // base: element (x=0,k=0) of this workgroup's operand panel; sx/sk element strides along x / k;
  // xlim/klim: number of valid x / k from `base` on (unused by the plain VEC modes).
  __device__ __forceinline__ void load(const E *__restrict__ base, int64_t sx, int64_t sk,
                                       int64_t k0, int64_t xlim, int64_t klim, int t,
                                       const GemmArgs<E> *cg = nullptr) {
#pragma unroll
    for (int i = 0; i < NV; i++) load_op(base, sx, sk, k0, xlim, klim, t, i, cg);
  }

  __device__ __forceinline__ void store(E *__restrict__ lds, int t) const {
#pragma unroll
    for (int gi = 0; gi < NV / GROUPN; gi++)
#pragma unroll
      for (int c = 0; c < WOPS; c++) store_op(lds, t, gi, c);
  }

  // The same work cut into single-instruction "ops" so the main loop can slot one op between two
  // MFMAs instead of issuing the whole staging block at once (which idles the matrix pipe):
  //   piece i (one 16-B register vector):  WOPS LDS-write ops (EPV scalar writes when transposing,
  //   1 x ds_write_b128 otherwise), then 1 load op that refills the vector for the tile after next.
  //   (pair mode: an op group is 2 pieces: 4 x ds_write_b64, then the 2 loads.)
  static constexpr int WOPS = KQ ? (ALONG_K ? 2 : 4) : (ALONG_K ? EPV : 1);  // LDS-write ops per group
  static constexpr int OPS_PER_GROUP = WOPS + GROUPN;
  static constexpr int NOPS = (NV / GROUPN) * OPS_PER_GROUP;

  __device__ __forceinline__ E masked(int i, int c) const {
    if constexpr (MASKED)
      return ((msk[i] >> c) & 1u) ? v[i][c] : (E)0;
    else
      return v[i][c];
  }

  // op c of group gi (a group is one piece, or a pair of pieces in pair mode)
  __device__ __forceinline__ void store_op(E *__restrict__ lds, int t, int gi, int c) const {
    const int i = gi;  // (single-piece groups)
    const int idx = t + i * NT;
    if constexpr (PATCH) {
      *reinterpret_cast<Vec *>(lds + pl_dst[i]) = v[i];
      return;
    }
    if constexpr (PAIRS) {
      // element c of pieces 2gi (k) and 2gi + 1 (k + 2): adjacent words of chunk 2J + (k & 1) in row x = 4xq + c
      int xq, k;
      piece_xk(t, 2 * gi, xq, k);
      const int x = 4 * xq + c;
      const int chunk = (2 * (k / 8) + (k & 1)) ^ kq_swz<BK>(x);
      typedef E E2 __attribute__((ext_vector_type(2)));
      E2 w;
      w[0] = masked(2 * gi, c);
      w[1] = masked(2 * gi + 1, c);
      *reinterpret_cast<E2 *>(lds + kq_row(x) * BK + 4 * chunk + ((k % 8) >> 1)) = w;
    } else if constexpr (!ALONG_K) {
      int xq, k;
      piece_xk(t, i, xq, k);
      Vec q = v[i];
      if constexpr (MASKED) {
#pragma unroll
        for (int e = 0; e < EPV; e++) q[e] = masked(i, e);
      }
      *reinterpret_cast<Vec *>(lds + k * BX + ((EPV * xq) ^ swz<E, BK>(k))) = q;
    } else if constexpr (KQ) {
      // op c = MFMA half: elements (c, c + 2) of the piece are k = 4kq + c and 4kq + c + 2, adjacent in chunk 2J + c
      const int kq = idx % (BK / 4), x = idx / (BK / 4);
      const int chunk = (2 * (kq / 2) + c) ^ kq_swz<BK>(x);
      typedef E E2 __attribute__((ext_vector_type(2)));
      E2 w;
      w[0] = masked(i, c);
      w[1] = masked(i, c + 2);
      *reinterpret_cast<E2 *>(lds + kq_row(x) * BK + 4 * chunk + 2 * (kq % 2)) = w;
    } else {
      const int kq = idx % (BK / EPV), x = idx / (BK / EPV);
      const int xs = x ^ swz<E, BK>(EPV * kq);
      lds[(EPV * kq + c) * BX + xs] = masked(i, c);
    }
  }

  __device__ __forceinline__ void load_op(const E *__restrict__ base, int64_t sx, int64_t sk, int64_t k0,
                                          int64_t xlim, int64_t klim, int t, int i,
                                          const GemmArgs<E> *cg = nullptr) {
    const int idx = t + i * NT;
    constexpr uint32_t ALL = (1u << EPV) - 1u;
    if constexpr (PATCH) {
      // one 16-byte piece of the patch for the K-tile whose first channel is p_c0 (pieces are called in order
      // i = 0 .. NV-1 once per tile: the last one advances the tile state by BK)
      const int khw = cg->ckH * cg->ckW, C = (int)(klim / khw);
      const int c = p_c0 + pg_ch[i];
      // invalid piece or channel beyond C -> offset -1 (sign bits OR-ed in: a select here becomes an exec-masked
      // region with a branch in the middle of the steady-state loop)
      const int off = (c * (cg->cH * cg->cW) + pg_off[i]) | (pg_off[i] >> 31) | ((C - 1 - c) >> 31);
      v[i] = buf_load((uint32_t)(off << 2));
      if (i == NV - 1) {
        // BK = cdc*kH*kW + cdr*kW + cdq (launcher); arithmetic carry, not a branch (uniform ifs become s_cbranch
        // and cut the steady-state loop into basic blocks)
        const int r = p_rem + cg->cdr * cg->ckW + cg->cdq;
        const int w = (r >= khw) ? 1 : 0;
        p_rem = r - w * khw;
        p_c0 = p_c0 + cg->cdc + w;
      }
      return;
    }
    if constexpr (CONV) {
      // gather the 4 pixels of this piece for k = (kc_, kr_, kq_), then advance k by BK
      const int c = kc_[i], kr = kr_[i], kq = kq_[i];
      const int koff = c * (cg->cH * cg->cW) + kr * cg->cW + kq;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t w = bad[i][e >> 1];
        // v_bfe_i32 of one bit: 0 (exists) or -1 (padding)
        const int inv = __builtin_amdgcn_sbfe(w, 16 * (e & 1) + kr, 1) | __builtin_amdgcn_sbfe(w, 16 * (e & 1) + 8 + kq, 1);
        const int voff = ((org[i][e] + koff) | inv) << 2;
        v[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
      }
      int nq = kq + cg->cdq, nr = kr + cg->cdr, nc = c + cg->cdc;
      if (nq >= cg->ckW) { nq -= cg->ckW; nr++; }
      if (nr >= cg->ckH) { nr -= cg->ckH; nc++; }
      kq_[i] = nq; kr_[i] = nr; kc_[i] = nc;
    } else if constexpr (!ALONG_K) {
      int xq, k;
      piece_xk(t, i, xq, k);
      if constexpr (VEC && !EDGE) {
        v[i] = *reinterpret_cast<const Vec *>(base + (k0 + k) * sk + EPV * xq);
      } else {
        const int64_t kk = k0 + k;
        const bool kin = kk < klim;
        const int64_t kc_ = kin ? kk : klim - 1;
        if constexpr (EDGE) {
          // x beyond xlim only feeds outputs that are never stored; k beyond klim is masked below (and lies
          // past the operand's end anyway).  32-bit offsets: the dispatcher keeps the panel under 4 GB.
          v[i] = buf_load((uint32_t)((kk * sk + EPV * xq) * (int64_t)sizeof(E)));
        } else {
          const E *p = base + kc_ * sk;
#pragma unroll
          for (int c = 0; c < EPV; c++) {
            const int64_t x = EPV * xq + c;
            v[i][c] = p[(x < xlim ? x : xlim - 1) * sx];
          }
        }
        msk[i] = kin ? ALL : 0u;
      }
    } else {
      const int kq = idx % (BK / EPV), x = idx / (BK / EPV);
      if constexpr (VEC && !EDGE) {
        v[i] = *reinterpret_cast<const Vec *>(base + (int64_t)x * sx + k0 + EPV * kq);
      } else {
        const int64_t kk = k0 + EPV * kq;
        const int64_t xc = (x < xlim) ? x : xlim - 1;
        if constexpr (EDGE) {
          // the k tail must read as zero (the next elements in memory belong to the next row): per-element mask
          v[i] = buf_load((uint32_t)(((int64_t)x * sx + kk) * (int64_t)sizeof(E)));
          const int64_t left = klim - kk;
          msk[i] = left >= EPV ? ALL : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
        } else {
          const E *p = base + xc * sx;
          uint32_t m = 0;
#pragma unroll
          for (int c = 0; c < EPV; c++) {
            const int64_t kc_ = kk + c;
            v[i][c] = p[(kc_ < klim ? kc_ : klim - 1) * sk];
            m |= (kc_ < klim ? 1u : 0u) << c;
          }
          msk[i] = m;
        }
      }
    }
  }
};

// ---- the kernel -----------------------------------------------------------------------------------
// STAGES = 2: double-buffered LDS, one barrier at the end of every K-tile (fill next stage, barrier).
// STAGES = 3: LDS ring with the barrier in the MIDDLE of the K-tile's MFMA stream:
//     iteration t:  R (tile t+1, loaded during t-1) -> LDS[(t+1)%3] ; issue HBM loads of tile t+2 -> R ;
//                   first half of tile t's MFMAs ; barrier ; second half ; prefetch tile t+1's first
//                   fragments.  The stage written in iteration t was last read in iteration t-2, and
//                   every wave has passed barrier t-1 => finished t-2: no WAR race; reads of tile t+1
//                   start only after barrier t => after every wave's stores: no RAW race.  The
//                   vmcnt -> ds_write -> lgkmcnt -> barrier -> ds_read chain that idles the matrix
//                   pipe at every tile boundary of the 2-stage form disappears.
// Fragments are register double-buffered (k-step j+1 is read from LDS while step j's MFMAs issue).
//
// __launch_bounds__ 2nd argument = waves per SIMD the register allocator must leave room for.
// A1: the launcher saw alpha == 1 (the only value any reference caller uses): 1*x is x bit for bit, so the laser-order
// slice fold drops its multiply -- 16 scalar adds per accumulator block between two MFMAs instead of 32 operations.
template <typename E, int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool EXACT, int STAGES, int OCC,
          bool DBG = false, bool EPI = false, bool A1 = false>
__global__ void __launch_bounds__(WM *WN * 64, OCC)
    gemm_mfma_kernel(const GemmArgs<E> g) {
  using M_ = Mma<E>;
  using Acc = typename M_::Acc;
  constexpr int MB = M_::MB, KS = M_::KS, ACC = M_::ACC;
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / MB, TN = WTN / MB;
  constexpr int NJ = BK / KS;  // MFMA k-steps per K-tile
  static_assert(WTM % MB == 0 && WTN % MB == 0, "wave tile must be built from whole MFMA blocks");
  static_assert(STAGES == 2 || STAGES == 3, "2 or 3 LDS stages");
  static_assert(STAGES == 2 || NJ >= 4, "ring form prefetches past the mid-tile barrier");
  constexpr int STAGE = BK * (BM + BN);  // elements per LDS stage: A panel then B panel

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  E *const smem = reinterpret_cast<E *>(smem_raw);

  // -- which C tile: XCD-aware (bijective) remap, then a grouped raster (GROUP_M tile-rows per group) --
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, xcd = bid % 8, loc = bid / 8, q = nwg / 8, r = nwg % 8;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // the 32 workgroups resident on an XCD form a GROUP_M x (32/GROUP_M) patch of tiles; its HBM/MALL traffic per
  // K-tile is ~ (GROUP_M*BM + 32/GROUP_M*BN), smallest for a square patch: 8x4 of square tiles, 4x8 of 256x128
  constexpr int GROUP_M = (BM >= 2 * BN) ? 4 : 8;
  const int width = GROUP_M * g.tiles_n;
  const int group = wgid / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int pid_m = first_m + (wgid % width) % gsz;
  const int pid_n = (wgid % width) / gsz;
  const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN + g.col0;
  const int64_t bz = blockIdx.y;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, lo = M_::lx(lane), hi = M_::lk(lane);
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;

  const E *Ab = g.A + bz * g.bsA + m0 * g.rsA;  // x = row of A, k along csA
  // x = col of B, k along rsB; for the implicit-GEMM conv the "matrix" is the NCHW image itself
  constexpr bool BPATCH = (BMODE == LOAD_CONV_PATCH);
  constexpr bool BCONV = (BMODE == LOAD_IM2COL) || BPATCH;
  // K-slice-parallel conv (GemmArgs::cs_imgs): z -> (image, kc slice); this workgroup covers K-tiles [kt0, nkt).
  // cs_len is a multiple of 2*BK, so kt0 is even (the 2-stage form's stage parity is kt & 1).
  int kt0 = 0;
  int64_t kend = g.K, bimg = bz;
  if constexpr (BCONV) {
    if (g.cs_imgs > 0) {
      const int slice = (int)(bz / g.cs_imgs);
      bimg = bz - (int64_t)slice * g.cs_imgs;
      kt0 = slice * (g.cs_len / BK);
      kend = min(g.K, (int64_t)(slice + 1) * g.cs_len);
    }
  }
  const E *Bb = BCONV ? g.B + bimg * g.bsB : g.B + bz * g.bsB + n0 * g.csB;
  E *Cb = g.C + bz * g.bsC;
  const int64_t K = g.K;
  const int64_t mlim = g.M - m0, nlim = g.N - n0;

  // k-quad LDS image policy (sweep_f32_v10/v11.json, 8192^3): taken whenever BOTH operands can use it --
  // k-contiguous pieces (2 x ds_write_b64), or x-contiguous pieces in pair mode (an even number of pieces per
  // thread: 4 x ds_write_b64 per pair).  B transposed: +1.3..2 %; plain row-major A and B: 256x256 fast
  // 138.9 -> 141.5, 256x128x32 fast 133.1 -> 136.1 / laser-order 132.8 -> 133.8, 128x128 fast 134 -> 136.4.
  // Never for one operand alone (k-quad A + k-major B: 256x128 laser-order -1.5 %); single transposed pieces
  // (8 x ds_write_b32 per 2 pieces) cost the 256x256 tile 4 %, hence pairs.
  constexpr bool A_K = (AMODE == LOAD_VEC_K || AMODE == LOAD_GEN_K || AMODE == LOAD_VEC_K_EDGE);
  constexpr bool B_K = (BMODE == LOAD_VEC_K || BMODE == LOAD_GEN_K || BMODE == LOAD_VEC_K_EDGE);
  constexpr bool A_X = (AMODE == LOAD_VEC_X || AMODE == LOAD_VEC_X_EDGE);
  // (the im2col gather could feed the image in pair mode as well: measured neutral on C4, left on the k-major image)
  constexpr bool B_X = (BMODE == LOAD_VEC_X || BMODE == LOAD_VEC_X_EDGE);
  constexpr bool A_XP = A_X && ((BM * BK / 4) / NT) % 2 == 0, B_XP = B_X && ((BN * BK / 4) / NT) % 2 == 0;  // pair mode possible
  constexpr bool KQ_ON = LH_KQ && STAGES == 3 && std::is_same<E, float>::value &&
                         (A_K || A_XP) && (B_K || B_XP);
  // (a k-quad filter operand next to the patch-gathered B of the conv: measured 1 % slower on C4 -- both or none)
  constexpr bool KQA = KQ_ON, KQB = KQ_ON;
  TileLoader<E, BM, BK, NT, AMODE, KQA> la;
  TileLoader<E, BN, BK, NT, BMODE, KQB> lb;
  // (EDGE modes) operand panels as bounds-checked buffers: span = offset of the panel's last valid element + 1
  la.init_buf(Ab, (g.Mext - m0 - 1) * g.rsA + (g.Kext - 1) * g.csA + 1);
  lb.init_buf(Bb, (g.Next - n0 - 1) * g.csB + (g.Kext - 1) * g.rsB + 1);
  lb.init_conv(g, n0, t, Bb, kt0 * BK);
  lb.init_patch(g, n0, t, Bb, kt0 * BK);
  // LOAD_CONV_PATCH: per B block of this wave, where the lane's output pixel sits in the patch (float index of its
  // window origin in channel slot 0), and the running k position of the fragment reads (uniform)
  int pbase[BPATCH ? TN : 1];
  if constexpr (BPATCH) {
    const int row0 = (int)(n0 / g.coW) * g.csH - g.cpH;
#pragma unroll
    for (int n = 0; n < TN; n++) {
      int64_t x = n0 + wn0 + MB * n + lo;
      if (x >= g.N) x = g.N - 1;  // columns beyond N feed outputs that are never stored
      const int oh = (int)(x / g.coW), ow = (int)(x - (int64_t)oh * g.coW);
      pbase[n] = (oh * g.csH - g.cpH - row0) * g.cPWs + ow * g.csW - g.cpW + 4;
    }
    // one-time clear of the B regions of all stages: padding columns / never-loaded cells must read as zero
    for (int o = t * 4; o < BK * BN; o += NT * 4)
#pragma unroll
      for (int st_ = 0; st_ < STAGES; st_++) {
        typedef E E4z __attribute__((ext_vector_type(4)));
        *reinterpret_cast<E4z *>(smem + st_ * STAGE + BK * BM + o) = E4z{};
      }
    __syncthreads();
  }

  Acc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < ACC; r++) acc[i][n][r] = (E)0;

  const E alpha = g.alpha, beta = g.beta;

  // C element owned by (block i, block n, accumulator element r) of this lane
  auto c_ptr = [&](int i, int n, int r, bool &ok) __attribute__((always_inline)) -> E * {
    const int64_t row = m0 + wm0 + MB * i + M_::acc_row(r, lane);
    const int64_t col = n0 + wn0 + MB * n + M_::acc_col(lane);
    ok = (row < g.M) && (col < g.N);
    return Cb + row * g.rsC + col * g.csC;
  };
  // beta*C0 exactly as the reference's epilogues do it: beta == 0 -> 0 without reading C,
  // beta == 1 -> C, else C*beta (one rounding)  [gemm_ukernel_generic.nim:59-66, 107-115]
  auto scaled_c0 = [&](int i, int n, int r) __attribute__((always_inline)) -> E {
    if (beta == (E)0) return (E)0;
    bool ok;
    const E *p = c_ptr(i, n, r, ok);
    const E c0 = ok ? *p : (E)0;
    return beta == (E)1 ? c0 : M_::mul(c0, beta);
  };
  // C += AB or C += alpha*AB, unfused  [gemm_ukernel_generic.nim:68-76]
  // (alpha == 1 needs no special case: 1*x is x bit for bit, so the multiply is always issued)
  auto axpy = [&](E run, E ab) __attribute__((always_inline)) -> E { return M_::add(run, M_::mul(alpha, ab)); };

  Acc run[EXACT ? TM : 1][EXACT ? TN : 1];
  if constexpr (EXACT) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int n = 0; n < TN; n++)
#pragma unroll
        for (int r = 0; r < ACC; r++) run[i][n][r] = scaled_c0(i, n, r);
  }

  const int nkt = (int)((kend + BK - 1) / BK);
  const int kc_tiles = EXACT ? (g.kc / BK) : 0;

  // fragment of k-step j: lane feeds k = KS*j + lk(lane) (the MFMA consumes them in ascending order,
  // continuing the ascending-k chain), x = block base + lx(lane)
  // Fragments are read and consumed in GROUPS of KGRP consecutive k-steps so that every pinned group holds
  // ~8 MFMAs whatever the wave tile: with 4 MFMAs per group (64x64 wave tile, KGRP = 1) the bare
  // MFMA + fragment-read stream already ran 5 % slower than with 8 (ablation, scripts/ablate_f32.py).
  // Ring of 2 group slots (group g+1 is read from LDS while group g's MFMAs issue); NG is even, so
  // slot = g & 1 stays compile-time across tiles.
  // (the 2-stage form has no cross-tile fragment prefetch: the first group's LDS latency is exposed once
  // per tile, so larger groups only lengthen the exposed part -- measured 127 -> 111 TFLOP/s at KGRP = 4)
  constexpr int KGRP_WANT = 8 / (TM * TN) < 1 ? 1 : 8 / (TM * TN);
  // a k-quad operand hands over 4 k-steps per ds_read_b128: groups of 4 k-steps
  constexpr int KGRP = (KQA || KQB) ? 4 : STAGES == 2 ? 1 : (KGRP_WANT > NJ / 4 ? (NJ / 4 < 1 ? 1 : NJ / 4) : KGRP_WANT);
  constexpr int NG = NJ / KGRP;  // groups per K-tile
  static_assert(NJ % KGRP == 0 && NG % 2 == 0 && NG >= 2, "BK must give an even number of fragment groups");
  E fa[2][KGRP][TM], fb[2][KGRP][TN];
  const int kqs = (KQA || KQB) ? kq_swz<(BK == 16 || BK == 32) ? BK : 16>(lo) : 0;  // block bases are multiples of 32: swizzle of x = of lo
  const int lor = kq_row(lo);                                                       // likewise the physical row
  auto ldgroup = [&](const E *sA, const E *sB, int grp, int slot) __attribute__((always_inline)) {
    if constexpr (KQA || KQB) {
      typedef E E4 __attribute__((ext_vector_type(4)));
      const int chunk = 4 * ((2 * grp + hi) ^ kqs);  // the 16 bytes of this lane's MFMA half for k-steps 4grp .. 4grp+3
      if constexpr (KQA) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const E4 q = *reinterpret_cast<const E4 *>(sA + (wm0 + MB * i + lor) * BK + chunk);
#pragma unroll
          for (int u = 0; u < 4; u++) fa[slot][u][i] = q[u];
        }
      }
      if constexpr (KQB) {
#pragma unroll
        for (int n = 0; n < TN; n++) {
          const E4 q = *reinterpret_cast<const E4 *>(sB + (wn0 + MB * n + lor) * BK + chunk);
#pragma unroll
          for (int u = 0; u < 4; u++) fb[slot][u][n] = q[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KGRP; u++) {
      const int j = grp * KGRP + u;
      const int k = KS * j + hi;
      // all k of one step share a swizzle when the step fits one 16-byte piece (fp32); otherwise it is per lane
      const int s = (KS <= M_::EPV) ? swz<E, BK>(KS * j) : swz<E, BK>(k);
      if constexpr (!KQA) {
#pragma unroll
        for (int i = 0; i < TM; i++) fa[slot][u][i] = sA[k * BM + wm0 + MB * i + (lo ^ s)];
      }
      if constexpr (BPATCH) {
        // this lane's k of step j -> its view's start in the patch, from the tile's offset table ([hi][k-step])
        const int ko = reinterpret_cast<const int32_t *>(sB)[decltype(lb)::TBL + hi * (BK / 2) + j];
#pragma unroll
        for (int n = 0; n < TN; n++) fb[slot][u][n] = sB[pbase[BPATCH ? n : 0] + ko];
      } else if constexpr (!KQB) {
#pragma unroll
        for (int n = 0; n < TN; n++) fb[slot][u][n] = sB[k * BN + wn0 + MB * n + (lo ^ s)];
      }
    }
  };
  auto mfma_group = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < KGRP; u++)
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int n = 0; n < TN; n++) acc[i][n] = M_::mma(fa[slot][u][i], fb[slot][u][n], acc[i][n]);
  };
  // Laser's pc loop: the micro-kernel accumulator restarts at +0 for every kc slice and the slice sum
  // is added into C (gemm.nim:150-158; ukernel zero-init gemm_ukernel_generator.nim:189)
  auto fold = [&]() __attribute__((always_inline)) {
    if constexpr (EXACT) {
      asm volatile("; laser-order slice fold" ::: "memory");  // keeps this a real (rare) block, never if-converted
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int n = 0; n < TN; n++)
#pragma unroll
          for (int r = 0; r < ACC; r++) {
            run[i][n][r] = axpy(run[i][n][r], acc[i][n][r]);
            acc[i][n][r] = (E)0;
          }
    }
  };

  if constexpr (STAGES == 2) {
    // -- prologue: tile kt0 (0 but for the K-slice form) -> LDS stage 0 --
    la.load(Ab, g.rsA, g.csA, (int64_t)kt0 * BK, mlim, K, t);
    lb.load(Bb, g.csB, g.rsB, (int64_t)kt0 * BK, nlim, K, t, &g);
    la.store(smem, t);
    lb.store(smem + BK * BM, t);
    lb.store_table(smem + BK * BM, t, &g);
    __syncthreads();
    auto k_tile2 = [&](auto MORE_, int kt) __attribute__((always_inline)) {
      constexpr bool more = decltype(MORE_)::value;
      const E *sA = smem + (kt & 1) * STAGE;
      const E *sB = sA + BK * BM;
      if (more) {  // issue the next tile's HBM loads before the MFMA block (latency hides under it)
        la.load(Ab, g.rsA, g.csA, (int64_t)(kt + 1) * BK, mlim, K, t);
        lb.load(Bb, g.csB, g.rsB, (int64_t)(kt + 1) * BK, nlim, K, t, &g);
      }
      ldgroup(sA, sB, 0, 0);
#pragma unroll
      for (int gi = 0; gi < NG; gi++) {
        if (gi + 1 < NG) ldgroup(sA, sB, gi + 1, (gi + 1) & 1);
        // pin the order "LDS reads of group g+1, then MFMAs of group g": without it hipcc sinks each
        // ds_read down to its first use and every MFMA group eats the full LDS latency
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(gi & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto refill = [&](int kt) __attribute__((always_inline)) {
      E *dA = smem + ((kt + 1) & 1) * STAGE;
      la.store(dA, t);
      lb.store(dA + BK * BM, t);
      lb.store_table(dA + BK * BM, t, &g);
      __syncthreads();
    };
    // slice folds live outside the steady-state loop (a test inside it is if-converted into
    // predicated VALU work on every tile)
    int kt = kt0;
    int next_fold = (EXACT && kc_tiles > 0) ? kc_tiles : 0x7fffffff;
    for (;;) {
      const int stop = min(next_fold - 1, nkt - 1);  // tiles [kt, stop) are followed by another tile of the same slice
      for (; kt < stop; kt++) {
        k_tile2(std::true_type{}, kt);
        refill(kt);
      }
      if (kt == next_fold - 1 && kt + 1 < nkt) {  // last tile of a slice, more slices follow
        k_tile2(std::true_type{}, kt);
        fold();
        refill(kt);
        kt++;
        next_fold += kc_tiles;
        continue;
      }
      break;
    }
    if (kt < nkt) k_tile2(std::false_type{}, kt);
  } else {
    // -- prologue: tile kt0 (0 but for the K-slice form) -> LDS stage 0; tile kt0+1 -> registers --
    la.load(Ab, g.rsA, g.csA, (int64_t)kt0 * BK, mlim, K, t);
    lb.load(Bb, g.csB, g.rsB, (int64_t)kt0 * BK, nlim, K, t, &g);
    la.store(smem, t);
    lb.store(smem + BK * BM, t);
    lb.store_table(smem + BK * BM, t, &g);
    if (nkt > kt0 + 1) {
      la.load(Ab, g.rsA, g.csA, (int64_t)(kt0 + 1) * BK, mlim, K, t);
      lb.load(Bb, g.csB, g.rsB, (int64_t)(kt0 + 1) * BK, nlim, K, t, &g);
    }
    __syncthreads();
    ldgroup(smem, smem + BK * BM, 0, 0);
    int st = 0;  // stage holding tile kt
    // One K-tile.  MORE / MORE2 (tile kt+1 / kt+2 exist) are compile-time so that the steady-state
    // body is ONE basic block: hipcc then derives exact counted `s_waitcnt vmcnt(N)` for the
    // interleaved loads; with run-time guards every staging op became its own block behind a
    // conservative vmcnt(0), i.e. a full HBM round trip per op.
    // FOLD_: this tile is the first of a Laser kc slice -- each accumulator block is folded into `run`
    // right before its first MFMA of the tile, and that MFMA takes a literal-zero C operand.  The fold's
    // VALU work (16 packed ops per block) then issues in the shadow of the previous block's MFMA instead
    // of stalling the matrix pipe for the whole fold at a slice boundary.
    auto k_tile = [&](auto MORE_, auto MORE2_, auto FOLD_, int kt) __attribute__((always_inline)) {
      constexpr bool more = decltype(MORE_)::value, more2 = decltype(MORE2_)::value;
      constexpr bool fold_here = EXACT && decltype(FOLD_)::value;
      const E *sA = smem + st * STAGE;
      const E *sB = sA + BK * BM;
      const int st1 = (st == 2) ? 0 : st + 1;
      const E *nA = smem + st1 * STAGE;
      const E *nB = nA + BK * BM;
      E *wA = smem + st1 * STAGE;
      E *wB = wA + BK * BM;
      const int64_t k2 = (int64_t)(kt + 2) * BK;
      // staging op `o` of this iteration: A pieces first, then B pieces; per piece its LDS writes
      // (tile kt+1, from registers) followed by the HBM load that refills the registers (tile kt+2)
      auto staging_op = [&](int o) __attribute__((always_inline)) {
        constexpr int NA = decltype(la)::NOPS, NB = decltype(lb)::NOPS;
        if (o < NA) {
          constexpr int P = decltype(la)::OPS_PER_GROUP, W = decltype(la)::WOPS, GN = decltype(la)::GROUPN;
          const int gi = o / P, c = o % P;
          if (c < W) {
            if (LH_DBG_KEEP(2)) la.store_op(wA, t, gi, c);
          } else if (more2 && LH_DBG_KEEP(1)) {
            la.load_op(Ab, g.rsA, g.csA, k2, mlim, K, t, gi * GN + (c - W));
          }
        } else if (o < NA + NB) {
          constexpr int P = decltype(lb)::OPS_PER_GROUP, W = decltype(lb)::WOPS, GN = decltype(lb)::GROUPN;
          const int gi = (o - NA) / P, c = (o - NA) % P;
          if (c < W) {
            if (LH_DBG_KEEP(2)) lb.store_op(wB, t, gi, c);
            if (gi == 0 && c == 0) lb.store_table(wB, t, &g);
          } else if (more2 && LH_DBG_KEEP(1)) {
            lb.load_op(Bb, g.csB, g.rsB, k2, nlim, K, t, gi * GN + (c - W), &g);
          }
        }
      };
      constexpr int NOPS = decltype(la)::NOPS + decltype(lb)::NOPS;
      constexpr int NMF = TM * TN;                     // MFMA slots per k-step
      constexpr int SLOTS = (NJ / 2) * NMF;            // slots before the mid-tile barrier
      constexpr int PER = (NOPS + SLOTS - 1) / SLOTS;  // staging ops per slot (1 unless the tile is tiny)
#pragma unroll
      for (int gi = 0; gi < NG; gi++) {
        // everyone's stores of tile kt+1 are done past this point
        if (gi == NG / 2 && LH_DBG_KEEP(4)) __syncthreads();
        if (gi + 1 < NG)
          ldgroup(sA, sB, gi + 1, (gi + 1) & 1);
        else if (more)
          ldgroup(nA, nB, 0, 0);  // first fragment group of the next tile (past the barrier)
        // pin the order "LDS reads of group g+1, then MFMAs of group g": without it hipcc sinks each
        // ds_read down to its first use and every MFMA group eats the full LDS latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KGRP; u++)
#pragma unroll
          for (int i = 0; i < TM; i++)
#pragma unroll
            for (int n = 0; n < TN; n++) {
              if (fold_here && gi == 0 && u == 0) {
                if constexpr (A1)
                  run[EXACT ? i : 0][EXACT ? n : 0] = M_::fold_acc1(run[EXACT ? i : 0][EXACT ? n : 0], acc[i][n]);
                else
                  run[EXACT ? i : 0][EXACT ? n : 0] = M_::fold_acc(run[EXACT ? i : 0][EXACT ? n : 0], acc[i][n], alpha);
                acc[i][n] = M_::mma(fa[0][0][i], fb[0][0][n], Acc{});
              } else {
                acc[i][n] = M_::mma(fa[gi & 1][u][i], fb[gi & 1][u][n], acc[i][n]);
              }
              if (more && gi < NG / 2) {
                const int slot = (gi * KGRP + u) * NMF + i * TN + n;
#pragma unroll
                for (int q = 0; q < PER; q++) staging_op(slot * PER + q);
              }
              __builtin_amdgcn_sched_barrier(0);  // one staging op rides behind each MFMA
            }
      }
      st = st1;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // Slice boundaries are handled OUTSIDE the steady-state loop: a fold test inside it gets
    // if-converted by hipcc into 256 predicated VALU ops per K-tile (measured -3 %).
    int kt = kt0;
    int next_fold = (EXACT && kc_tiles > 0) ? kc_tiles : 0x7fffffff;  // fold once tiles [.., next_fold) are done
    for (;;) {
      const int stop = min(next_fold, nkt - 2);
      for (; kt < stop; kt++) k_tile(T_{}, T_{}, F_{}, kt);
      if (kt == next_fold && kt < nkt - 2) {  // slice boundary in the steady state: the fold rides inside the tile
        k_tile(T_{}, T_{}, T_{}, kt);
        kt++;
        next_fold += kc_tiles;
        continue;
      }
      break;
    }
    // the last two tiles (no more HBM loads / no more LDS writes); a slice boundary here folds in the open
    if (kt == next_fold && kt < nkt) {
      fold();
      next_fold += kc_tiles;
    }
    if (kt + 1 < nkt) {
      k_tile(T_{}, F_{}, F_{}, kt);
      kt++;
      if (kt == next_fold && kt < nkt) {
        fold();
        next_fold += kc_tiles;
      }
    }
    if (kt < nkt) k_tile(F_{}, F_{}, F_{}, kt);
  }

  // -- epilogue: last (or only) slice, then store with the caller's strides --
  // optional fused tail (uniform switch, outside the hot loop): + bias (one rounding), then the activation
  // (EPI is a template flag: the plain kernels carry none of this -- the inlined tanh/exp bodies, one per
  // accumulator element, triple the code size)
  auto epilogue = [&](E v, int i, int n, int r, bool ok) __attribute__((always_inline)) -> E {
    if (g.bias != nullptr) {
      const int64_t row = m0 + wm0 + MB * i + M_::acc_row(r, lane);
      const int64_t col = n0 + wn0 + MB * n + M_::acc_col(lane);
      const E b = ok ? g.bias[bz * g.bsBias + row * g.rsBias + col * g.csBias] : (E)0;
      v = M_::add(v, b);
    }
    switch (g.act) {
      case 1: v = v > (E)0 ? v : (E)0; break;                          // relu (NaN -> 0, like max(x, 0) on the host)
      case 2: v = M_::tanh_(v); break;
      case 3: v = (E)1 / ((E)1 + M_::exp_(-v)); break;                 // sigmoid
      default: break;
    }
    return v;
  };
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < ACC; r++) {
        bool ok;
        E *p = c_ptr(i, n, r, ok);
        E base;
        if constexpr (EXACT)
          base = run[i][n][r];
        else
          base = scaled_c0(i, n, r);
        E out = axpy(base, acc[i][n][r]);
        if constexpr (EPI) out = epilogue(out, i, n, r, ok);
        if (ok) *p = out;
      }
}

// ---- per-configuration launcher ---------------------------------------------------------------------
template <typename E, int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool EXACT, int STAGES, int OCC,
          bool DBG = false, bool EPI = false, bool A1 = false>
hipError_t launch_one(const GemmArgs<E> &a, hipStream_t s) {
  auto kern = gemm_mfma_kernel<E, BM, BN, BK, WM, WN, AMODE, BMODE, EXACT, STAGES, OCC, DBG, EPI, A1>;
  constexpr size_t lds = (size_t)STAGES * BK * (BM + BN) * sizeof(E);
  static_assert(lds <= 160 * 1024, "LDS budget is 160 KiB per CU");
  static PerDeviceOnce attr;  // one per instantiation
  if (hipError_t e = attr.run([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      });
      e != hipSuccess)
    return e;
  GemmArgs<E> g = a;
  g.tiles_m = (int)((a.M + BM - 1) / BM);
  g.tiles_n = (int)((a.N - a.col0 + BN - 1) / BN);
  if constexpr (BMODE == LOAD_CONV_PATCH) {
    const int khw = a.ckH * a.ckW;
    ConvPatchGeom pg;
    if (!conv_patch_geom(BK, BN, a.cW, a.coW, a.ckH, a.ckW, a.csH, a.csW, &pg) || a.cW % 4 != 0) return hipErrorInvalidValue;
    g.cdc = BK / khw;
    g.cRp = pg.rp;
    g.cPWs = pg.pws;
    g.cCHM = pg.chm;
    g.cdr = (BK % khw) / a.ckW;
    g.cdq = (BK % khw) % a.ckW;
  }
  if constexpr (BMODE == LOAD_IM2COL) {
    const int khw = a.ckH * a.ckW;
    g.cdc = BK / khw;
    g.cdr = (BK % khw) / a.ckW;
    g.cdq = (BK % khw) % a.ckW;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)a.batch, 1), block(WM * WN * 64, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, s, g);
  return hipGetLastError();
}

// dispatch over the loader modes for one tile configuration
template <typename E, int BM, int BN, int BK, int WM, int WN, int STAGES, int OCC, bool WITH_VEC, bool WITH_GEN, bool EXACT>
hipError_t launch_cfg_mode(const GemmArgs<E> &a, int amode, int bmode, hipStream_t s) {
  const bool fused = a.bias != nullptr || a.act != 0;  // fused epilogue: separate instantiation
  // alpha == 1 variant of the laser-order kernels (plain epilogue, float32 only: the headline path)
  const bool a1 = EXACT && std::is_same<E, float>::value && !fused && a.alpha == (E)1;
  (void)a1;
#define LH_CASE(AM, BMD) \
  if (amode == AM && bmode == BMD) {                                                                          \
    if constexpr (EXACT && std::is_same<E, float>::value) {                                                    \
      if (a1) return launch_one<E, BM, BN, BK, WM, WN, AM, BMD, EXACT, STAGES, OCC, false, false, true>(a, s);  \
    }                                                                                                          \
    return fused ? launch_one<E, BM, BN, BK, WM, WN, AM, BMD, EXACT, STAGES, OCC, false, true>(a, s)            \
                 : launch_one<E, BM, BN, BK, WM, WN, AM, BMD, EXACT, STAGES, OCC, false, false>(a, s);          \
  }
  if constexpr (WITH_VEC) {
    LH_CASE(LOAD_VEC_K, LOAD_VEC_X)
    LH_CASE(LOAD_VEC_K, LOAD_VEC_K)
    LH_CASE(LOAD_VEC_X, LOAD_VEC_X)
    LH_CASE(LOAD_VEC_X, LOAD_VEC_K)
    LH_CASE(LOAD_VEC_K_EDGE, LOAD_VEC_X_EDGE)
    LH_CASE(LOAD_VEC_K_EDGE, LOAD_VEC_K_EDGE)
    LH_CASE(LOAD_VEC_X_EDGE, LOAD_VEC_X_EDGE)
    LH_CASE(LOAD_VEC_X_EDGE, LOAD_VEC_K_EDGE)
    // (the laser-order 256x128x32 kernel has no registers left for the gather state -- the conv launcher maps it
    // to the BK = 16 form of the same tile -- so its conv instantiations are not built at all)
    constexpr bool CONV_OK = !(EXACT && BM == 256 && BN == 128 && BK == 32);
    if constexpr (std::is_same<E, float>::value && CONV_OK) {
      LH_CASE(LOAD_VEC_K, LOAD_IM2COL)  // implicit-GEMM conv: filter [C_out][C_in*kH*kW] is k-contiguous
      LH_CASE(LOAD_VEC_K_EDGE, LOAD_IM2COL)
      LH_CASE(LOAD_VEC_K, LOAD_CONV_PATCH)
      LH_CASE(LOAD_VEC_K_EDGE, LOAD_CONV_PATCH)
    }
  }
  if constexpr (WITH_GEN) {
    LH_CASE(LOAD_GEN_K, LOAD_GEN_X)
    // (both operands through the scalar k-walking loaders hold the most address state: at 3 waves per SIMD -- 168
    // registers -- that pair spilled 2 VGPRs on the 128x128 tile; it is built for 2)
    if constexpr (OCC > 2) {
      if (amode == LOAD_GEN_K && bmode == LOAD_GEN_K)
        return fused ? launch_one<E, BM, BN, BK, WM, WN, LOAD_GEN_K, LOAD_GEN_K, EXACT, STAGES, 2, false, true>(a, s)
                     : launch_one<E, BM, BN, BK, WM, WN, LOAD_GEN_K, LOAD_GEN_K, EXACT, STAGES, 2, false, false>(a, s);
    } else {
      LH_CASE(LOAD_GEN_K, LOAD_GEN_K)
    }
    LH_CASE(LOAD_GEN_X, LOAD_GEN_X)
    LH_CASE(LOAD_GEN_X, LOAD_GEN_K)
    if constexpr (std::is_same<E, float>::value) {
      LH_CASE(LOAD_GEN_K, LOAD_IM2COL)
    }
  }
#undef LH_CASE
  return hipErrorInvalidValue;
}

}  // namespace laser_hip
