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
  static constexpr bool ALONG_K = (MODE == LOAD_VEC_K || MODE == LOAD_GEN_K);
  static constexpr bool VEC = (MODE == LOAD_VEC_X || MODE == LOAD_VEC_K);
  f32x4 v[NV];

  // base: element (x=0,k=0) of this workgroup's operand panel; sx/sk element strides along x / k;
  // xlim/klim: number of valid x / k from `base` on (only used by the GEN modes).
  __device__ __forceinline__ void load(const float *__restrict__ base, int64_t sx, int64_t sk,
                                       int64_t k0, int64_t xlim, int64_t klim, int t) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int idx = t + i * NT;
      if constexpr (!ALONG_K) {
        const int xq = idx % (BX / 4), k = idx / (BX / 4);
        if constexpr (VEC) {
          v[i] = *reinterpret_cast<const f32x4 *>(base + (k0 + k) * sk + 4 * xq);
        } else {
          const int64_t kk = k0 + k;
          const float *p = base + kk * sk + (int64_t)(4 * xq) * sx;
          const bool kin = kk < klim;
#pragma unroll
          for (int c = 0; c < 4; c++) v[i][c] = (kin && (4 * xq + c) < xlim) ? p[c * sx] : 0.0f;
        }
      } else {
        const int kq = idx % (BK / 4), x = idx / (BK / 4);
        if constexpr (VEC) {
          v[i] = *reinterpret_cast<const f32x4 *>(base + (int64_t)x * sx + k0 + 4 * kq);
        } else {
          const int64_t kk = k0 + 4 * kq;
          const float *p = base + (int64_t)x * sx + kk * sk;
          const bool xin = x < xlim;
#pragma unroll
          for (int c = 0; c < 4; c++) v[i][c] = (xin && (kk + c) < klim) ? p[c * sk] : 0.0f;
        }
      }
    }
  }

  __device__ __forceinline__ void store(float *__restrict__ lds, int t) const {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int idx = t + i * NT;
      if constexpr (!ALONG_K) {
        const int xq = idx % (BX / 4), k = idx / (BX / 4);
        *reinterpret_cast<f32x4 *>(lds + k * BX + ((4 * xq) ^ swz<BK>(k))) = v[i];
      } else {
        const int kq = idx % (BK / 4), x = idx / (BK / 4);
        const int xs = x ^ swz<BK>(4 * kq);
#pragma unroll
        for (int c = 0; c < 4; c++) lds[(4 * kq + c) * BX + xs] = v[i][c];
      }
    }
  }
};

// ---- the kernel -----------------------------------------------------------------------------------
// __launch_bounds__ 2nd argument = waves per SIMD the register allocator must leave room for:
// 256-thread workgroups ask for 2-3 (several workgroups per CU share each SIMD, so one workgroup's
// MFMAs cover another's barrier / LDS-fill bubbles); 512-thread workgroups already put 2 on a SIMD.
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool EXACT>
__global__ void __launch_bounds__(WM *WN * 64, (WM * WN * 64 >= 512) ? 2 : (EXACT ? 2 : 3))
    gemm_f32_mfma_kernel(const GemmArgs<float> g) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be built from 32x32 MFMA blocks");
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
  const float *Bb = g.B + bz * g.bsB + n0 * g.csB;  // x = col of B, k along rsB
  float *Cb = g.C + bz * g.bsC;
  const int64_t K = g.K;
  const int64_t mlim = g.M - m0, nlim = g.N - n0;

  TileLoader<BM, BK, NT, AMODE> la;
  TileLoader<BN, BK, NT, BMODE> lb;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][n][r] = 0.0f;

  const float alpha = g.alpha, beta = g.beta;

  // C element owned by (i, n, r): row = wm0 + 32 i + (r&3) + 8 (r>>2) + 4 hi, col = wn0 + 32 n + lo
  auto c_ptr = [&](int i, int n, int r, bool &ok) -> float * {
    const int64_t row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int64_t col = n0 + wn0 + 32 * n + lo;
    ok = (row < g.M) && (col < g.N);
    return Cb + row * g.rsC + col * g.csC;
  };
  // beta*C0 exactly as the reference's epilogues do it: beta == 0 -> 0 without reading C,
  // beta == 1 -> C, else C*beta (one rounding)  [gemm_ukernel_generic.nim:59-66, 107-115]
  auto scaled_c0 = [&](int i, int n, int r) -> float {
    if (beta == 0.0f) return 0.0f;
    bool ok;
    const float *p = c_ptr(i, n, r, ok);
    const float c0 = ok ? *p : 0.0f;
    return beta == 1.0f ? c0 : __fmul_rn(c0, beta);
  };
  // C += AB or C += alpha*AB, unfused  [gemm_ukernel_generic.nim:68-76]
  auto axpy = [&](float run, float ab) -> float {
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

  // -- prologue: tile 0 -> LDS stage 0 --
  la.load(Ab, g.rsA, g.csA, 0, mlim, K, t);
  lb.load(Bb, g.csB, g.rsB, 0, nlim, K, t);
  la.store(smem, t);
  lb.store(smem + BK * BM, t);
  __syncthreads();

  int until_fold = kc_tiles;
  for (int kt = 0; kt < nkt; kt++) {
    const float *sA = smem + (kt & 1) * STAGE;
    const float *sB = sA + BK * BM;
    const bool more = (kt + 1) < nkt;
    if (more) {  // issue the next tile's HBM loads before the MFMA block (latency hides under it)
      la.load(Ab, g.rsA, g.csA, (int64_t)(kt + 1) * BK, mlim, K, t);
      lb.load(Bb, g.csB, g.rsB, (int64_t)(kt + 1) * BK, nlim, K, t);
    }

#pragma unroll
    for (int j = 0; j < BK / 2; j++) {
      const int k = 2 * j + hi;  // lanes 0-31 feed k=2j, lanes 32-63 feed k=2j+1: ascending-k chain
      const int s = swz<BK>(2 * j);
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = sA[k * BM + wm0 + 32 * i + (lo ^ s)];
#pragma unroll
      for (int n = 0; n < TN; n++) b[n] = sB[k * BN + wn0 + 32 * n + (lo ^ s)];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int n = 0; n < TN; n++)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[n], acc[i][n], 0, 0, 0);
    }

    if constexpr (EXACT) {
      // Laser's pc loop: the micro-kernel accumulator restarts at +0 for every kc slice and the
      // slice sum is added into C (gemm.nim:150-158; ukernel zero-init gemm_ukernel_generator.nim:189)
      if (--until_fold == 0 && more) {
        until_fold = kc_tiles;
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
    }

    if (more) {
      float *dA = smem + ((kt + 1) & 1) * STAGE;
      la.store(dA, t);
      lb.store(dA + BK * BM, t);
    }
    __syncthreads();
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
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool EXACT>
hipError_t launch_one(const GemmArgs<float> &a, hipStream_t s) {
  auto kern = gemm_f32_mfma_kernel<BM, BN, BK, WM, WN, AMODE, BMODE, EXACT>;
  constexpr size_t lds = 2 * BK * (BM + BN) * sizeof(float);
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
template <int BM, int BN, int BK, int WM, int WN, bool WITH_VEC, bool WITH_GEN, bool EXACT>
hipError_t launch_cfg_mode(const GemmArgs<float> &a, int amode, int bmode, hipStream_t s) {
#define LH_CASE(AM, BMD) \
  if (amode == AM && bmode == BMD) return launch_one<BM, BN, BK, WM, WN, AM, BMD, EXACT>(a, s);
  if constexpr (WITH_VEC) {
    LH_CASE(LOAD_VEC_K, LOAD_VEC_X)
    LH_CASE(LOAD_VEC_K, LOAD_VEC_K)
    LH_CASE(LOAD_VEC_X, LOAD_VEC_X)
    LH_CASE(LOAD_VEC_X, LOAD_VEC_K)
  }
  if constexpr (WITH_GEN) {
    LH_CASE(LOAD_GEN_K, LOAD_GEN_X)
    LH_CASE(LOAD_GEN_K, LOAD_GEN_K)
    LH_CASE(LOAD_GEN_X, LOAD_GEN_X)
    LH_CASE(LOAD_GEN_X, LOAD_GEN_K)
  }
#undef LH_CASE
  return hipErrorInvalidValue;
}

}  // namespace laser_hip
