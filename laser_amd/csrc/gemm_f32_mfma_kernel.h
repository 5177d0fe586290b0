// laser_amd/csrc/gemm_f32_mfma_kernel.h -- fp32 GEMM on the gfx950 exact-f32 matrix cores.
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
//   pc loop over kc=512 slices, C += per slice     K loop inside the workgroup, double-buffered LDS;
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
#pragma once
#include <type_traits>

#include "common.h"

namespace laser_hip {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BK>
struct SwzShift {
  static_assert(BK == 8 || BK == 16 || BK == 32, "BK must be 8, 16 or 32");
  static constexpr int value = (BK == 32) ? 2 : (BK == 16) ? 3 : 4;
};

template <int BK>
__device__ __forceinline__ int swz(int k) {
  return ((k >> 2) & 7) << SwzShift<BK>::value;
}

// ---- operand tile loader: HBM -> registers -> LDS panel (the "packing" stage) -------------------
template <int BX, int BK, int NT, int MODE>
struct TileLoader {
  static constexpr int NV = (BX * BK / 4) / NT;  // 16-B pieces per thread per tile
  static_assert(NV >= 1 && (BX * BK / 4) % NT == 0, "tile must split evenly over the workgroup");
  static constexpr bool ALONG_K = (MODE == LOAD_VEC_K || MODE == LOAD_GEN_K || MODE == LOAD_VEC_K_EDGE);
  static constexpr bool EDGE = (MODE == LOAD_VEC_X_EDGE || MODE == LOAD_VEC_K_EDGE);
  static constexpr bool VEC = (MODE == LOAD_VEC_X || MODE == LOAD_VEC_K || EDGE);
  static constexpr bool CONV = (MODE == LOAD_IM2COL);
  f32x4 v[NV];
  // LOAD_IM2COL: per piece and element, (oh*sH - pH) in the high and (ow*sW - pW) in the low 16 bits
  // of the output pixel this lane gathers for (fixed for the whole K loop); 0x7fff7fff = beyond N.
  int32_t pix[CONV ? NV : 1][CONV ? 4 : 1];

  __device__ __forceinline__ void init_conv(const GemmArgs<float> &g, int64_t n0, int t) {
    if constexpr (CONV) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int idx = t + i * NT;
        const int xq = idx % (BX / 4);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int64_t j = n0 + 4 * xq + e;
          if (j < g.N) {
            const int oh = (int)(j / g.coW), ow = (int)(j - (int64_t)oh * g.coW);
            const int r0 = oh * g.csH - g.cpH, c0 = ow * g.csW - g.cpW;
            pix[i][e] = (int32_t)(((uint32_t)(r0 & 0xffff) << 16) | (uint32_t)(c0 & 0xffff));
          } else {
            pix[i][e] = 0x7fff7fff;  // row 32767: fails the bounds test for every kr
          }
        }
      }
    }
  }

  // base: element (x=0,k=0) of this workgroup's operand panel; sx/sk element strides along x / k;
  // xlim/klim: number of valid x / k from `base` on (only used by the GEN modes).
  __device__ __forceinline__ void load(const float *__restrict__ base, int64_t sx, int64_t sk,
                                       int64_t k0, int64_t xlim, int64_t klim, int t,
                                       const GemmArgs<float> *cg = nullptr) {
#pragma unroll
    for (int i = 0; i < NV; i++) load_op(base, sx, sk, k0, xlim, klim, t, i, cg);
  }

  __device__ __forceinline__ void store(float *__restrict__ lds, int t) const {
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
      for (int c = 0; c < WOPS; c++) store_op(lds, t, i, c);
  }

  // The same work cut into single-instruction "ops" so the main loop can slot one op between two
  // MFMAs instead of issuing the whole staging block at once (which idles the matrix pipe):
  //   piece i (one 16-B register quad):  WOPS LDS-write ops (4 x ds_write_b32 when transposing,
  //   1 x ds_write_b128 otherwise), then 1 load op that refills the quad for the tile after next.
  static constexpr int WOPS = ALONG_K ? 4 : 1;
  static constexpr int OPS_PER_PIECE = WOPS + 1;
  static constexpr int NOPS = NV * OPS_PER_PIECE;

  __device__ __forceinline__ void store_op(float *__restrict__ lds, int t, int i, int c) const {
    const int idx = t + i * NT;
    if constexpr (!ALONG_K) {
      const int xq = idx % (BX / 4), k = idx / (BX / 4);
      *reinterpret_cast<f32x4 *>(lds + k * BX + ((4 * xq) ^ swz<BK>(k))) = v[i];
    } else {
      const int kq = idx % (BK / 4), x = idx / (BK / 4);
      const int xs = x ^ swz<BK>(4 * kq);
      lds[(4 * kq + c) * BX + xs] = v[i][c];
    }
  }

  __device__ __forceinline__ void load_op(const float *__restrict__ base, int64_t sx, int64_t sk, int64_t k0,
                                          int64_t xlim, int64_t klim, int t, int i,
                                          const GemmArgs<float> *cg = nullptr) {
    const int idx = t + i * NT;
    if constexpr (CONV) {
      // k -> (channel, kernel row, kernel col); the pixel part was decoded once in init_conv
      const int k = idx / (BX / 4);
      const int kk = (int)k0 + k;
      const int khw = cg->ckH * cg->ckW;
      const int c = kk / khw, rem = kk - c * khw;
      const int kr = rem / cg->ckW, kc = rem - kr * cg->ckW;
      const bool kin = kk < (int)klim;
      const float *img = base + (int64_t)c * cg->cH * cg->cW;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int row = (int)(int16_t)(pix[i][e] >> 16) + kr, col = (int)(int16_t)(pix[i][e] & 0xffff) + kc;
        const bool ok = kin && (unsigned)row < (unsigned)cg->cH && (unsigned)col < (unsigned)cg->cW;
        v[i][e] = ok ? img[row * cg->cW + col] : 0.0f;
      }
    } else if constexpr (!ALONG_K) {
      const int xq = idx % (BX / 4), k = idx / (BX / 4);
      if constexpr (VEC && !EDGE) {
        v[i] = *reinterpret_cast<const f32x4 *>(base + (k0 + k) * sk + 4 * xq);
      } else if constexpr (EDGE) {
        // xlim % 4 == 0 (checked by the dispatcher): a quad is entirely inside or outside in x
        const int64_t kk = k0 + k;
        const bool kin = kk < klim;
        const int64_t xc = (4 * xq < xlim) ? 4 * xq : xlim - 4;
        const f32x4 q = *reinterpret_cast<const f32x4 *>(base + (kin ? kk : 0) * sk + xc);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        v[i] = kin ? q : z;
      } else {
        const int64_t kk = k0 + k;
        const float *p = base + kk * sk + (int64_t)(4 * xq) * sx;
        const bool kin = kk < klim;
#pragma unroll
        for (int c = 0; c < 4; c++) v[i][c] = (kin && (4 * xq + c) < xlim) ? p[c * sx] : 0.0f;
      }
    } else {
      const int kq = idx % (BK / 4), x = idx / (BK / 4);
      if constexpr (VEC && !EDGE) {
        v[i] = *reinterpret_cast<const f32x4 *>(base + (int64_t)x * sx + k0 + 4 * kq);
      } else if constexpr (EDGE) {
        // klim % 4 == 0 (checked by the dispatcher): a quad is entirely inside or outside in k
        const int64_t kk = k0 + 4 * kq;
        const bool kin = kk < klim;
        const int64_t xc = (x < xlim) ? x : xlim - 1;
        const f32x4 q = *reinterpret_cast<const f32x4 *>(base + xc * sx + (kin ? kk : 0));
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        v[i] = kin ? q : z;
      } else {
        const int64_t kk = k0 + 4 * kq;
        const float *p = base + (int64_t)x * sx + kk * sk;
        const bool xin = x < xlim;
#pragma unroll
        for (int c = 0; c < 4; c++) v[i][c] = (xin && (kk + c) < klim) ? p[c * sk] : 0.0f;
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
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool EXACT, int STAGES, int OCC,
          bool DBG = false>
__global__ void __launch_bounds__(WM *WN * 64, OCC)
    gemm_f32_mfma_kernel(const GemmArgs<float> g) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int NJ = BK / 2;  // MFMA k-steps (k-pairs) per K-tile
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be built from 32x32 MFMA blocks");
  static_assert(STAGES == 2 || STAGES == 3, "2 or 3 LDS stages");
  static_assert(STAGES == 2 || BK >= 16, "ring form prefetches past the mid-tile barrier: needs NJ/2 >= 2");
  constexpr int STAGE = BK * (BM + BN);  // floats per LDS stage: A panel then B panel

  extern __shared__ __attribute__((aligned(16))) float smem[];

  // -- which C tile: XCD-aware (bijective) remap, then a grouped raster (8 tile-rows per group) --
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, xcd = bid % 8, loc = bid / 8, q = nwg / 8, r = nwg % 8;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * g.tiles_n;
  const int group = wgid / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int pid_m = first_m + (wgid % width) % gsz;
  const int pid_n = (wgid % width) / gsz;
  const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN;
  const int64_t bz = blockIdx.y;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, lo = lane & 31, hi = lane >> 5;
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;

  const float *Ab = g.A + bz * g.bsA + m0 * g.rsA;  // x = row of A, k along csA
  // x = col of B, k along rsB; for the implicit-GEMM conv the "matrix" is the NCHW image itself
  const float *Bb = (BMODE == LOAD_IM2COL) ? g.B + bz * g.bsB : g.B + bz * g.bsB + n0 * g.csB;
  float *Cb = g.C + bz * g.bsC;
  const int64_t K = g.K;
  const int64_t mlim = g.M - m0, nlim = g.N - n0;

  TileLoader<BM, BK, NT, AMODE> la;
  TileLoader<BN, BK, NT, BMODE> lb;
  lb.init_conv(g, n0, t);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][n][r] = 0.0f;

  const float alpha = g.alpha, beta = g.beta;

  // C element owned by (i, n, r): row = wm0 + 32 i + (r&3) + 8 (r>>2) + 4 hi, col = wn0 + 32 n + lo
  auto c_ptr = [&](int i, int n, int r, bool &ok) __attribute__((always_inline)) -> float * {
    const int64_t row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int64_t col = n0 + wn0 + 32 * n + lo;
    ok = (row < g.M) && (col < g.N);
    return Cb + row * g.rsC + col * g.csC;
  };
  // beta*C0 exactly as the reference's epilogues do it: beta == 0 -> 0 without reading C,
  // beta == 1 -> C, else C*beta (one rounding)  [gemm_ukernel_generic.nim:59-66, 107-115]
  auto scaled_c0 = [&](int i, int n, int r) __attribute__((always_inline)) -> float {
    if (beta == 0.0f) return 0.0f;
    bool ok;
    const float *p = c_ptr(i, n, r, ok);
    const float c0 = ok ? *p : 0.0f;
    return beta == 1.0f ? c0 : __fmul_rn(c0, beta);
  };
  // C += AB or C += alpha*AB, unfused  [gemm_ukernel_generic.nim:68-76]
  auto axpy = [&](float run, float ab) __attribute__((always_inline)) -> float {
    return __fadd_rn(run, alpha == 1.0f ? ab : __fmul_rn(alpha, ab));
  };

  f32x16 run[EXACT ? TM : 1][EXACT ? TN : 1];
  if constexpr (EXACT) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int n = 0; n < TN; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) run[i][n][r] = scaled_c0(i, n, r);
  }

  const int nkt = (int)((K + BK - 1) / BK);
  const int kc_tiles = EXACT ? (g.kc / BK) : 0;

  // fragment of k-step j: lanes 0-31 feed k = 2j, lanes 32-63 feed k = 2j+1 -> ascending-k chain
  // fragment ring: 4 slots (NJ is a multiple of 4, so slot = j & 3 stays compile-time across tiles);
  // the 3-stage loop reads PFD steps ahead, the 2-stage loop one step ahead
  float fa[4][TM], fb[4][TN];
  constexpr int PFD = 1;  // 2 was measured: no gain at 1 WG/CU and it costs the 128-VGPR occupancy step of the fast 256x128 kernel
  static_assert(NJ % 4 == 0, "BK must be a multiple of 8");
  auto ldfrag = [&](const float *sA, const float *sB, int j, int slot) __attribute__((always_inline)) {
    const int k = 2 * j + hi;
    const int s = swz<BK>(2 * j);
#pragma unroll
    for (int i = 0; i < TM; i++) fa[slot][i] = sA[k * BM + wm0 + 32 * i + (lo ^ s)];
#pragma unroll
    for (int n = 0; n < TN; n++) fb[slot][n] = sB[k * BN + wn0 + 32 * n + (lo ^ s)];
  };
  auto mfma_step = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int n = 0; n < TN; n++)
        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i], fb[slot][n], acc[i][n], 0, 0, 0);
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
          for (int r = 0; r < 16; r++) {
            run[i][n][r] = axpy(run[i][n][r], acc[i][n][r]);
            acc[i][n][r] = 0.0f;
          }
    }
  };

  if constexpr (STAGES == 2) {
    // -- prologue: tile 0 -> LDS stage 0 --
    la.load(Ab, g.rsA, g.csA, 0, mlim, K, t);
    lb.load(Bb, g.csB, g.rsB, 0, nlim, K, t, &g);
    la.store(smem, t);
    lb.store(smem + BK * BM, t);
    __syncthreads();
    auto k_tile2 = [&](auto MORE_, int kt) __attribute__((always_inline)) {
      constexpr bool more = decltype(MORE_)::value;
      const float *sA = smem + (kt & 1) * STAGE;
      const float *sB = sA + BK * BM;
      if (more) {  // issue the next tile's HBM loads before the MFMA block (latency hides under it)
        la.load(Ab, g.rsA, g.csA, (int64_t)(kt + 1) * BK, mlim, K, t);
        lb.load(Bb, g.csB, g.rsB, (int64_t)(kt + 1) * BK, nlim, K, t, &g);
      }
      ldfrag(sA, sB, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        if (j + 1 < NJ) ldfrag(sA, sB, j + 1, (j + 1) & 3);
        // pin the order "LDS reads of step j+1, then MFMAs of step j": without it hipcc sinks each
        // ds_read down to its first use and every MFMA group eats the full LDS latency
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(j & 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto refill = [&](int kt) __attribute__((always_inline)) {
      float *dA = smem + ((kt + 1) & 1) * STAGE;
      la.store(dA, t);
      lb.store(dA + BK * BM, t);
      __syncthreads();
    };
    // slice folds live outside the steady-state loop (a test inside it is if-converted into
    // predicated VALU work on every tile)
    int kt = 0;
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
    // -- prologue: tile 0 -> LDS stage 0; tile 1 -> registers --
    la.load(Ab, g.rsA, g.csA, 0, mlim, K, t);
    lb.load(Bb, g.csB, g.rsB, 0, nlim, K, t, &g);
    la.store(smem, t);
    lb.store(smem + BK * BM, t);
    if (nkt > 1) {
      la.load(Ab, g.rsA, g.csA, BK, mlim, K, t);
      lb.load(Bb, g.csB, g.rsB, BK, nlim, K, t, &g);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PFD; j++) ldfrag(smem, smem + BK * BM, j, j);
    int st = 0;  // stage holding tile kt
    // One K-tile.  MORE / MORE2 (tile kt+1 / kt+2 exist) are compile-time so that the steady-state
    // body is ONE basic block: hipcc then derives exact counted `s_waitcnt vmcnt(N)` for the
    // interleaved loads; with run-time guards every staging op became its own block behind a
    // conservative vmcnt(0), i.e. a full HBM round trip per op.
    auto k_tile = [&](auto MORE_, auto MORE2_, int kt) __attribute__((always_inline)) {
      constexpr bool more = decltype(MORE_)::value, more2 = decltype(MORE2_)::value;
      const float *sA = smem + st * STAGE;
      const float *sB = sA + BK * BM;
      const int st1 = (st == 2) ? 0 : st + 1;
      const float *nA = smem + st1 * STAGE;
      const float *nB = nA + BK * BM;
      float *wA = smem + st1 * STAGE;
      float *wB = wA + BK * BM;
      const int64_t k2 = (int64_t)(kt + 2) * BK;
      // staging op `o` of this iteration: A pieces first, then B pieces; per piece its LDS writes
      // (tile kt+1, from registers) followed by the HBM load that refills the registers (tile kt+2)
      auto staging_op = [&](int o) __attribute__((always_inline)) {
        constexpr int NA = decltype(la)::NOPS, NB = decltype(lb)::NOPS;
        if (o < NA) {
          constexpr int P = decltype(la)::OPS_PER_PIECE;
          const int i = o / P, c = o % P;
          if (c < P - 1) {
            if (!DBG || !(g.dbg & 2)) la.store_op(wA, t, i, c);
          } else if (more2 && (!DBG || !(g.dbg & 1))) {
            la.load_op(Ab, g.rsA, g.csA, k2, mlim, K, t, i);
          }
        } else if (o < NA + NB) {
          constexpr int P = decltype(lb)::OPS_PER_PIECE;
          const int i = (o - NA) / P, c = (o - NA) % P;
          if (c < P - 1) {
            if (!DBG || !(g.dbg & 2)) lb.store_op(wB, t, i, c);
          } else if (more2 && (!DBG || !(g.dbg & 1))) {
            lb.load_op(Bb, g.csB, g.rsB, k2, nlim, K, t, i, &g);
          }
        }
      };
      constexpr int NOPS = decltype(la)::NOPS + decltype(lb)::NOPS;
      constexpr int NMF = TM * TN;                     // MFMA slots per k-step
      constexpr int SLOTS = (NJ / 2) * NMF;            // slots before the mid-tile barrier
      constexpr int PER = (NOPS + SLOTS - 1) / SLOTS;  // staging ops per slot (1 unless the tile is tiny)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        // everyone's stores of tile kt+1 are done past this point
        if (j == NJ / 2 && (!DBG || !(g.dbg & 4))) __syncthreads();
        if (j + PFD < NJ)
          ldfrag(sA, sB, j + PFD, (j + PFD) & 3);
        else if (more)
          ldfrag(nA, nB, j + PFD - NJ, (j + PFD) & 3);  // first fragments of the next tile (past the barrier)
        // pin the order "LDS reads of step j+PFD, then MFMAs of step j": without it hipcc sinks each
        // ds_read down to its first use and every MFMA group eats the full LDS latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int n = 0; n < TN; n++) {
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 3][i], fb[j & 3][n], acc[i][n], 0, 0, 0);
            if (more && j < NJ / 2) {
              const int slot = j * NMF + i * TN + n;
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
    int kt = 0;
    int next_fold = (EXACT && kc_tiles > 0) ? kc_tiles : 0x7fffffff;  // fold once tiles [.., next_fold) are done
    for (;;) {
      const int stop = min(next_fold, nkt - 2);
      for (; kt < stop; kt++) k_tile(T_{}, T_{}, kt);
      if (kt == next_fold && kt < nkt) {
        fold();
        next_fold += kc_tiles;
        continue;
      }
      break;
    }
    if (kt + 1 < nkt) {
      k_tile(T_{}, F_{}, kt);
      kt++;
      if (kt == next_fold && kt < nkt) {
        fold();
        next_fold += kc_tiles;
      }
    }
    if (kt < nkt) k_tile(F_{}, F_{}, kt);
  }

  // -- epilogue: last (or only) slice, then store with the caller's strides --
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        bool ok;
        float *p = c_ptr(i, n, r, ok);
        float base;
        if constexpr (EXACT)
          base = run[i][n][r];
        else
          base = scaled_c0(i, n, r);
        const float out = axpy(base, acc[i][n][r]);
        if (ok) *p = out;
      }
}

// ---- per-configuration launcher ---------------------------------------------------------------------
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool EXACT, int STAGES, int OCC,
          bool DBG = false>
hipError_t launch_one(const GemmArgs<float> &a, hipStream_t s) {
  auto kern = gemm_f32_mfma_kernel<BM, BN, BK, WM, WN, AMODE, BMODE, EXACT, STAGES, OCC, DBG>;
  constexpr size_t lds = (size_t)STAGES * BK * (BM + BN) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget is 160 KiB per CU");
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  GemmArgs<float> g = a;
  g.tiles_m = (int)((a.M + BM - 1) / BM);
  g.tiles_n = (int)((a.N + BN - 1) / BN);
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)a.batch, 1), block(WM * WN * 64, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, s, g);
  return hipGetLastError();
}

// dispatch over the loader modes for one tile configuration
template <int BM, int BN, int BK, int WM, int WN, int STAGES, int OCC, bool WITH_VEC, bool WITH_GEN, bool EXACT>
hipError_t launch_cfg_mode(const GemmArgs<float> &a, int amode, int bmode, hipStream_t s) {
#define LH_CASE(AM, BMD) \
  if (amode == AM && bmode == BMD) return launch_one<BM, BN, BK, WM, WN, AM, BMD, EXACT, STAGES, OCC>(a, s);
  if constexpr (WITH_VEC) {
    LH_CASE(LOAD_VEC_K, LOAD_VEC_X)
    LH_CASE(LOAD_VEC_K, LOAD_VEC_K)
    LH_CASE(LOAD_VEC_X, LOAD_VEC_X)
    LH_CASE(LOAD_VEC_X, LOAD_VEC_K)
    LH_CASE(LOAD_VEC_K_EDGE, LOAD_VEC_X_EDGE)
    LH_CASE(LOAD_VEC_K_EDGE, LOAD_VEC_K_EDGE)
    LH_CASE(LOAD_VEC_X_EDGE, LOAD_VEC_X_EDGE)
    LH_CASE(LOAD_VEC_X_EDGE, LOAD_VEC_K_EDGE)
    LH_CASE(LOAD_VEC_K, LOAD_IM2COL)       // implicit-GEMM conv: filter [C_out][C_in*kH*kW] is k-contiguous
    LH_CASE(LOAD_VEC_K_EDGE, LOAD_IM2COL)
  }
  if constexpr (WITH_GEN) {
    LH_CASE(LOAD_GEN_K, LOAD_GEN_X)
    LH_CASE(LOAD_GEN_K, LOAD_GEN_K)
    LH_CASE(LOAD_GEN_X, LOAD_GEN_X)
    LH_CASE(LOAD_GEN_X, LOAD_GEN_K)
    LH_CASE(LOAD_GEN_K, LOAD_IM2COL)
  }
#undef LH_CASE
  return hipErrorInvalidValue;
}

}  // namespace laser_hip
