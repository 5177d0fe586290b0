// laser_amd/csrc/limb_planes.h -- the packing pass of the integer limb GEMMs (gemm_i32_mfma.hip, gemm_i64_mfma.hip):
// a strided int32 / int64 operand -> NP = sizeof(T) int8 planes P_p[x][k] of balanced base-256 digits, k-contiguous,
// zero-padded to Xpad x Kpad.  The GPU analogue of pack_A_mc_kc / pack_B_kc_nc (gemm_packing.nim:24-94): strides are
// resolved here, B is transposed on the way.
//
// HBM-bound: algorithmic bytes = 2 * sizeof(T) * X * K (read the operand, write sizeof(T) one-byte planes).  A workgroup
// moves a 32 x 128 (x, k) tile through LDS so that BOTH sides coalesce whatever the source layout: the tile is read along
// the source's contiguous axis (k-contiguous: 128 k = 512 B / 1 KiB per row; x-contiguous: 32 x = 128 / 256 B per k), and
// written as 8 consecutive 16-byte chunks per (plane, x) row = 128 B segments.  (One thread per (x, 16 k) reading its own
// 16 elements straight from memory -- the first version -- ran at 1.2-2 TB/s: per-lane 64/128-byte strides on the
// k-contiguous side, 16-byte stores Kpad bytes apart on the x-contiguous side; profiles/r02/rocprof_int_limb.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace laser_hip {

constexpr int LP_TX = 32, LP_TK = 128, LP_THREADS = 256;

template <typename T>
struct LimbDigits;
template <>
struct LimbDigits<int32_t> {
  // bytes = balanced digits of a (add 128 to the three low bytes with carry propagation, flip their sign bits; the top
  // digit may be any representative mod 256 because 256^4 == 0 mod 2^32)
  static __device__ __forceinline__ void split(int32_t a, uint32_t w[1]) { w[0] = ((uint32_t)a + 0x00808080u) ^ 0x00808080u; }
};
template <>
struct LimbDigits<int64_t> {
  static __device__ __forceinline__ void split(int64_t a, uint32_t w[2]) {
    const uint64_t d = ((uint64_t)a + 0x0080808080808080ull) ^ 0x0080808080808080ull;
    w[0] = (uint32_t)d;
    w[1] = (uint32_t)(d >> 32);
  }
};

// planes[p][x][k] (int8), x < Xpad, k < Kpad (multiples of 32 / 128 are NOT required: edges are predicated);
// element (x, k) of the source at src[x*sx + k*sk] for x < X, k < K, zero elsewhere.  Kpad % 16 == 0.
template <typename T>
__global__ void __launch_bounds__(LP_THREADS) limb_planes_tiled_kernel(int8_t *__restrict__ planes, const T *__restrict__ src, int64_t X,
                                                                      int64_t K, int64_t sx, int64_t sk, int64_t Xpad, int64_t Kpad,
                                                                      int x_fast, int tiles_k, int tile_major, int vec) {
  constexpr int NW = (int)sizeof(T) / 4;  // 32-bit words per element = groups of four planes
  // [x][k] tile, rows padded by 4 elements (16 / 32 bytes): the phase-2 reads of 8 lanes walk one row, the next row starts
  // 16+ bytes further round the banks
  __shared__ __attribute__((aligned(16))) T tile[LP_TX][LP_TK + 4];
  const int t = threadIdx.x;
  const int64_t x0 = (int64_t)(blockIdx.x / tiles_k) * LP_TX, k0 = (int64_t)(blockIdx.x % tiles_k) * LP_TK;
  // phase 1: 4096 elements, 16 per thread, lanes along the source's contiguous axis.  vec (the launcher: unit stride along that axis,
  // the other stride and the base 16-byte aligned): 16-byte loads -- 4 / 8 per thread instead of 16 scalar ones; a vector that crosses
  // the operand's edge falls back to predicated elements
  typedef __attribute__((ext_vector_type(4))) int lp_vec16;
  constexpr int EV = 16 / (int)sizeof(T);
  union VecT { lp_vec16 q; T e[EV]; };
  if (vec && !x_fast) {
    constexpr int VPR = LP_TK / EV;
#pragma unroll
    for (int i = 0; i < LP_TX * VPR / LP_THREADS; i++) {
      const int v = t + i * LP_THREADS, xl = v / VPR, kl = (v % VPR) * EV;
      const int64_t x = x0 + xl, k = k0 + kl;
      VecT u;
      if (x < X && k + EV <= K) {
        u.q = *reinterpret_cast<const lp_vec16 *>(src + x * sx + k);
      } else {
#pragma unroll
        for (int j = 0; j < EV; j++) u.e[j] = (x < X && k + j < K) ? src[x * sx + k + j] : (T)0;
      }
      *reinterpret_cast<lp_vec16 *>(&tile[xl][kl]) = u.q;
    }
  } else if (vec) {
    constexpr int VPX = LP_TX / EV;
#pragma unroll
    for (int i = 0; i < LP_TK * VPX / LP_THREADS; i++) {
      const int v = t + i * LP_THREADS, kl = v / VPX, xl = (v % VPX) * EV;
      const int64_t x = x0 + xl, k = k0 + kl;
      VecT u;
      if (k < K && x + EV <= X) {
        u.q = *reinterpret_cast<const lp_vec16 *>(src + k * sk + x);
      } else {
#pragma unroll
        for (int j = 0; j < EV; j++) u.e[j] = (k < K && x + j < X) ? src[k * sk + x + j] : (T)0;
      }
#pragma unroll
      for (int j = 0; j < EV; j++) tile[xl + j][kl] = u.e[j];
    }
  } else
#pragma unroll
  for (int i = 0; i < LP_TX * LP_TK / LP_THREADS; i++) {
    const int e = t + i * LP_THREADS;
    const int xl = x_fast ? e % LP_TX : e / LP_TK;
    const int kl = x_fast ? e / LP_TX : e % LP_TK;
    const int64_t x = x0 + xl, k = k0 + kl;
    tile[xl][kl] = (x < X && k < K) ? src[x * sx + k * sk] : (T)0;
  }
  __syncthreads();
  // phase 2: thread -> (x = t / 8, 16-k chunk = t % 8): one 16-byte store per plane, 8 lanes = 128 contiguous bytes
  const int xl = t >> 3, kc = t & 7;
  const int64_t x = x0 + xl, kq = (k0 >> 4) + kc;
  if (x >= Xpad || kq * 16 >= Kpad) return;
  uint32_t out[4 * NW][4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    uint32_t w[4][NW];
#pragma unroll
    for (int c = 0; c < 4; c++) LimbDigits<T>::split(tile[xl][kc * 16 + g * 4 + c], w[c]);
    // 4x4 byte transposes: out[p][g] = { digit p of the 4 consecutive k }
#pragma unroll
    for (int h = 0; h < NW; h++) {
      const uint32_t lo01 = __builtin_amdgcn_perm(w[1][h], w[0][h], 0x05010400u), hi01 = __builtin_amdgcn_perm(w[1][h], w[0][h], 0x07030602u);
      const uint32_t lo23 = __builtin_amdgcn_perm(w[3][h], w[2][h], 0x05010400u), hi23 = __builtin_amdgcn_perm(w[3][h], w[2][h], 0x07030602u);
      out[4 * h + 0][g] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
      out[4 * h + 1][g] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
      out[4 * h + 2][g] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
      out[4 * h + 3][g] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
    }
  }
  typedef __attribute__((ext_vector_type(4))) int lp_i32x4;
  const int64_t plane = Xpad * Kpad;
  // tile_major = TR > 0 (the hand-scheduled kernels of laser_amd/asmgen/i8_kernel.py: TR = 128 rows for int32, 64 for int64;
  // Xpad % TR == 0, Kpad % 32 == 0): for every TR-row tile and 32-k tile one contiguous 16-KiB block [plane][k half][row][16
  // bytes] -- the global -> LDS stage of the GEMM is then a lane-linear copy and a fragment read is 32 consecutive chunks
  const int TR = tile_major > 0 ? tile_major : 1;
  const int64_t tbase = ((x / TR) * (Kpad >> 5) + (kq >> 1)) * (int64_t)(4 * NW * 2 * TR * 16) + (kq & 1) * (TR * 16) + (x % TR) * 16;
#pragma unroll
  for (int p = 0; p < 4 * NW; p++) {
    const lp_i32x4 q = {(int)out[p][0], (int)out[p][1], (int)out[p][2], (int)out[p][3]};
    if (tile_major)
      *reinterpret_cast<lp_i32x4 *>(planes + tbase + (int64_t)p * (2 * TR * 16)) = q;
    else
      *reinterpret_cast<lp_i32x4 *>(planes + p * plane + x * Kpad + kq * 16) = q;
  }
}

// The x-contiguous operand (a row-major B: element (x, k) at src[k * sk + x]) for the tile-major layout, 16-byte loads: a workgroup moves a
// 128 x 32 (x, k) tile -- 512-byte / 1-KiB row segments on the read side (the 32-x tile above reads 128 bytes per row), every thread
// transposes EV x EV blocks in registers (EV = elements per 16 bytes) so that LDS is written 16 bytes at a time, and phase 2 runs with
// lanes along x: a wave's store is 1 KiB contiguous in a block's [plane][k half][row][16 bytes] image.  Requires vec (launcher).
constexpr int LPX_TX = 128, LPX_TK = 32;
template <typename T>
__global__ void __launch_bounds__(LP_THREADS) limb_planes_xfast_kernel(int8_t *__restrict__ planes, const T *__restrict__ src, int64_t X, int64_t K,
                                                                      int64_t sk, int64_t Xpad, int64_t Kpad, int tiles_x, int TR) {
  constexpr int NW = (int)sizeof(T) / 4;
  constexpr int EV = 16 / (int)sizeof(T);
  typedef __attribute__((ext_vector_type(4))) int lp_vec16;
  union VecT { lp_vec16 q; T e[EV]; };
  __shared__ __attribute__((aligned(16))) T tile[LPX_TX][LPX_TK + 4];      // row stride 36 elements: phase 2's 8-lane groups are conflict-free
  const int t = threadIdx.x;
  // consecutive workgroups walk along x: neighbours read neighbouring 512-byte segments of the same 32 source rows
  const int64_t x0 = (int64_t)(blockIdx.x % tiles_x) * LPX_TX, k0 = (int64_t)(blockIdx.x / tiles_x) * LPX_TK;
  constexpr int XV = LPX_TX / EV, KB = LPX_TK / EV;      // blocks of EV x EV elements: XV along x (lanes), KB along k
#pragma unroll
  for (int i = 0; i < XV * KB / LP_THREADS; i++) {
    const int b = t + i * LP_THREADS, xl = (b % XV) * EV, kl = (b / XV) * EV;
    const int64_t x = x0 + xl, k = k0 + kl;
    VecT u[EV];
#pragma unroll
    for (int c = 0; c < EV; c++) {
      if (k + c < K && x + EV <= X) {
        u[c].q = *reinterpret_cast<const lp_vec16 *>(src + (k + c) * sk + x);
      } else {
#pragma unroll
        for (int j = 0; j < EV; j++) u[c].e[j] = (k + c < K && x + j < X) ? src[(k + c) * sk + x + j] : (T)0;
      }
    }
#pragma unroll
    for (int j = 0; j < EV; j++) {
      VecT w;
#pragma unroll
      for (int c = 0; c < EV; c++) w.e[c] = u[c].e[j];
      *reinterpret_cast<lp_vec16 *>(&tile[xl + j][kl]) = w.q;
    }
  }
  __syncthreads();
  const int xl = t & (LPX_TX - 1), kc = t >> 7;      // 128 x, two 16-k chunks
  const int64_t x = x0 + xl, kq = (k0 >> 4) + kc;
  if (x >= Xpad || kq * 16 >= Kpad) return;
  const int64_t tbase = ((x / TR) * (Kpad >> 5) + (kq >> 1)) * (int64_t)(4 * NW * 2 * TR * 16) + (kq & 1) * (TR * 16) + (x % TR) * 16;
#pragma unroll
  for (int h = 0; h < NW; h++) {
    uint32_t out[4][4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      uint32_t w[4][NW];
#pragma unroll
      for (int c = 0; c < 4; c++) LimbDigits<T>::split(tile[xl][kc * 16 + g * 4 + c], w[c]);
      const uint32_t lo01 = __builtin_amdgcn_perm(w[1][h], w[0][h], 0x05010400u), hi01 = __builtin_amdgcn_perm(w[1][h], w[0][h], 0x07030602u);
      const uint32_t lo23 = __builtin_amdgcn_perm(w[3][h], w[2][h], 0x05010400u), hi23 = __builtin_amdgcn_perm(w[3][h], w[2][h], 0x07030602u);
      out[0][g] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
      out[1][g] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
      out[2][g] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
      out[3][g] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const lp_vec16 q = {(int)out[p][0], (int)out[p][1], (int)out[p][2], (int)out[p][3]};
      *reinterpret_cast<lp_vec16 *>(planes + tbase + (int64_t)(4 * h + p) * (2 * TR * 16)) = q;
    }
  }
}

template <typename T>
inline hipError_t launch_limb_planes(int8_t *dst, const T *src, int64_t X, int64_t K, int64_t sx, int64_t sk, int64_t Xpad, int64_t Kpad,
                                     hipStream_t s, int tile_major = 0) {
  const int64_t tiles_x = (Xpad + LP_TX - 1) / LP_TX, tiles_k = (Kpad + LP_TK - 1) / LP_TK;
  if (tiles_x * tiles_k > 0x7fffffffll) return hipErrorInvalidValue;
  const int x_fast = (sx < 0 ? -sx : sx) < (sk < 0 ? -sk : sk);
  const int64_t ev = 16 / (int64_t)sizeof(T);
  const int vec = ((uintptr_t)src % 16 == 0) && (x_fast ? (sx == 1 && sk > 0 && sk % ev == 0) : (sk == 1 && sx > 0 && sx % ev == 0));
  if (vec && x_fast && tile_major > 0 && Kpad % 32 == 0) {
    const int64_t tx = (Xpad + LPX_TX - 1) / LPX_TX, tk = Kpad / LPX_TK;
    if (tx * tk > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(limb_planes_xfast_kernel<T>, dim3((unsigned)(tx * tk)), dim3(LP_THREADS), 0, s, dst, src, X, K, sk, Xpad, Kpad, (int)tx, tile_major);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(limb_planes_tiled_kernel<T>, dim3((unsigned)(tiles_x * tiles_k)), dim3(LP_THREADS), 0, s, dst, src, X, K, sx, sk, Xpad,
                     Kpad, x_fast, (int)tiles_k, tile_major, vec);
  return hipGetLastError();
}

}  // namespace laser_hip
