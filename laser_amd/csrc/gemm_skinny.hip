// laser_amd/csrc/gemm_skinny.hip -- gemm_strided when one output dimension is tiny (M <= 8 or N <= 8: matrix-vector
// products and their close relatives), all four element types.
//
// A 64x64 MFMA tile is 1/64 full on an N = 1 problem and the grid is a few dozen workgroups: the tiled kernels reach
// ~1 TB/s on what is a pure HBM stream (8192 x 1 x 8192: 0.36 ms).  Here the problem is viewed as
//     out[l][s] = sum_k X(l, k) * Y(k, s),   l < L (the long side),  s < S <= 8
// (N small: X = A, Y = B;  M small: X = B^T, Y = A^T, out = C^T -- only strides change), and one lane owns one
// (row l, kc slice p): it streams 2 KiB of its row from HBM and runs S accumulation chains.  The arithmetic is Laser's, restated per output
// element exactly as in the tiled kernels (gemm_ukernel_generator.nim:245-248, gemm.nim:150-158):
//     S_p = fma chain over the kc slice p from +0, ascending k;   C = (..((beta*C0 or 0) + alpha*S_0) + alpha*S_1 ..)
// and a v_fma_f32 / v_fma_f64 chain is bit-for-bit what the matrix cores compute (same fused multiply-add, same
// order), so results are identical to the tiled path and to the CPU restatement used by the tests (FAST mode takes the same path: Laser's own
// order is trivially inside its tolerance).
//
// Memory: the S columns of Y for the slices in flight sit in LDS (<= 64 KiB, read as broadcasts); X is read straight
// from HBM -- 16 bytes per lane per load when X is k-contiguous (each lane a private 2-KiB run: every 128-byte line is
// consumed over 8 consecutive loads out of L1), one coalesced dword per lane per k when X is l-contiguous.
#include <type_traits>

#include "common.h"

namespace laser_hip {

namespace {

constexpr int SK_SMAX = 8;

template <typename E>
struct SkinnyArgs {
  int64_t L, K;
  int32_t S, kc;  // kc == 0: one chain over all K (FAST)
  E alpha, beta;
  const E *X;
  int64_t sxl, sxk;
  const E *Y;
  int64_t syk, sys;
  E *C;
  int64_t scl, scs;
};

template <typename E>
__device__ __forceinline__ E fma_(E a, E b, E c);
template <>
__device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <>
__device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// integers: two's-complement wrap-around c + a*b (gemm_ukernel_avx2.nim:10-11), done on the unsigned type
template <>
__device__ __forceinline__ int32_t fma_<int32_t>(int32_t a, int32_t b, int32_t c) { return (int32_t)((uint32_t)a * (uint32_t)b + (uint32_t)c); }
template <>
__device__ __forceinline__ int64_t fma_<int64_t>(int64_t a, int64_t b, int64_t c) { return (int64_t)((uint64_t)a * (uint64_t)b + (uint64_t)c); }

template <typename E>
__device__ __forceinline__ E mul_(E a, E b) {
  if constexpr (std::is_integral<E>::value) {
    using U = typename std::make_unsigned<E>::type;
    return (E)((U)a * (U)b);
  } else {
#pragma clang fp contract(off)
    return a * b;
  }
}
template <typename E>
__device__ __forceinline__ E add_(E a, E b) {
  if constexpr (std::is_integral<E>::value) {
    using U = typename std::make_unsigned<E>::type;
    return (E)((U)a + (U)b);
  } else {
#pragma clang fp contract(off)
    return a + b;
  }
}

// Laser's kc slices are INDEPENDENT chains (each starts from +0); only their sums are added in order.  So the
// parallelism is (row l) x (slice p): lane (r, pl) of a workgroup computes S_p for its row over its 2 KiB of k, the
// partial sums meet in LDS and lane (r, 0) adds them in ascending p -- bit-identical to one lane walking all of K, with
// ceil(K / kc) times the lanes (and bytes in flight) of a row-per-lane kernel.
// XK: X is k-contiguous (sxk == 1) -> 16-byte loads along k; else one element per k (coalesced over l when sxl == 1)
template <typename E, int S, bool XK>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(const SkinnyArgs<E> g) {
  constexpr int EPV = 16 / sizeof(E);
  constexpr int KC = 2048 / sizeof(E);               // Laser's kc: 512 f32, 256 f64 (gemm_tiling.nim:310)
  constexpr int PB = S <= 2 ? 16 : S <= 4 ? 8 : 4;   // slices in flight per workgroup (Y for them: <= 64 KiB of LDS)
  constexpr int RB = 256 / PB;                       // rows per workgroup
  __shared__ E ys[PB * KC * S];
  __shared__ E sp[RB * PB * S];
  // lanes run along k slices when X is k-contiguous (each lane a private contiguous run), along l when X is
  // l-contiguous (64 consecutive rows = one coalesced 256-byte access per k)
  const int t = threadIdx.x, pl = XK ? t % PB : t / RB, r = XK ? t / PB : t % RB;
  const int64_t l = (int64_t)blockIdx.x * RB + r;
  const bool live = l < g.L;
  const E *xrow = g.X + (live ? l : g.L - 1) * g.sxl;  // dead rows re-read the last row (never stored)
  const bool vec_ok = XK && ((reinterpret_cast<uintptr_t>(xrow) & 15) == 0);
  const int64_t nsl = (g.K + KC - 1) / KC;

  E run[S];
#pragma unroll
  for (int s = 0; s < S; s++) {
    // beta*C0 exactly as the tiled epilogue does it: 0 without reading C, C0, or C0*beta (one rounding)
    E c0 = (E)0;
    if (g.beta != (E)0 && live && pl == 0) {
      c0 = g.C[l * g.scl + s * g.scs];
      if (g.beta != (E)1) c0 = mul_(c0, g.beta);
    }
    run[s] = c0;
  }
  for (int64_t p0 = 0; p0 < nsl; p0 += PB) {
    const int64_t kbase = p0 * KC;
    const int kspan = (int)(g.K - kbase < (int64_t)PB * KC ? g.K - kbase : (int64_t)PB * KC);
    __syncthreads();  // everyone is done with the previous group's Y and partial sums
    for (int i = t; i < kspan * S; i += 256) {
      const int k = i / S, s = i - k * S;
      ys[i] = g.Y[(kbase + k) * g.syk + s * g.sys];
    }
    __syncthreads();
    const int64_t p = p0 + pl;
    E acc[S];
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = (E)0;
    if (p < nsl) {
      const int64_t k0 = p * KC;
      const int kn = (int)(g.K - k0 < KC ? g.K - k0 : KC);
      const E *yp = ys + pl * KC * S;
      int k = 0;
      if (vec_ok) {
        typedef E EV __attribute__((ext_vector_type(EPV)));
        for (; k + 8 * EPV <= kn; k += 8 * EPV) {  // 8 independent 16-byte loads in flight per lane
          EV q[8];
#pragma unroll
          for (int u = 0; u < 8; u++) q[u] = *reinterpret_cast<const EV *>(xrow + k0 + k + u * EPV);
#pragma unroll
          for (int u = 0; u < 8; u++)
#pragma unroll
            for (int e = 0; e < EPV; e++)
#pragma unroll
              for (int s = 0; s < S; s++) acc[s] = fma_(q[u][e], yp[(k + u * EPV + e) * S + s], acc[s]);
        }
      } else {
        for (; k + 8 <= kn; k += 8) {
          E q[8];
#pragma unroll
          for (int u = 0; u < 8; u++) q[u] = xrow[(k0 + k + u) * g.sxk];
#pragma unroll
          for (int u = 0; u < 8; u++)
#pragma unroll
            for (int s = 0; s < S; s++) acc[s] = fma_(q[u], yp[(k + u) * S + s], acc[s]);
        }
      }
      for (; k < kn; k++) {
        const E x = xrow[(k0 + k) * g.sxk];
#pragma unroll
        for (int s = 0; s < S; s++) acc[s] = fma_(x, yp[k * S + s], acc[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < S; s++) sp[(r * PB + pl) * S + s] = acc[s];
    __syncthreads();
    if (pl == 0) {  // C += alpha * S_p, slice after slice (gemm.nim:150-158)
      const int cnt = (int)(nsl - p0 < PB ? nsl - p0 : PB);
      for (int q = 0; q < cnt; q++)
#pragma unroll
        for (int s = 0; s < S; s++) run[s] = add_(run[s], mul_(g.alpha, sp[(r * PB + q) * S + s]));
    }
  }
  if (live && pl == 0) {
#pragma unroll
    for (int s = 0; s < S; s++) g.C[l * g.scl + s * g.scs] = run[s];
  }
}

template <typename E, bool XK>
hipError_t launch_s(const SkinnyArgs<E> &a, hipStream_t s) {
  switch (a.S) {
#define LH_S(N)                                                                                        \
  case N: {                                                                                            \
    constexpr int RB = 256 / (N <= 2 ? 16 : N <= 4 ? 8 : 4);                                           \
    hipLaunchKernelGGL((gemm_skinny_kernel<E, N, XK>), dim3((unsigned)((a.L + RB - 1) / RB)), dim3(256), 0, s, a); \
  } break;
    LH_S(1) LH_S(2) LH_S(3) LH_S(4) LH_S(5) LH_S(6) LH_S(7) LH_S(8)
#undef LH_S
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

// Takes the problem when it is skinny enough to be an HBM stream; returns hipErrorNotSupported otherwise.
template <typename E>
hipError_t launch_gemm_skinny(const GemmArgs<E> &g, bool laser_order, int kc_elems, hipStream_t s) {
  if (g.batch != 1 || g.bias != nullptr || g.act != 0) return hipErrorNotSupported;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hipSuccess;
  const bool n_small = g.N <= SK_SMAX && g.M >= 512, m_small = g.M <= SK_SMAX && g.N >= 512;
  if (!n_small && !m_small) return hipErrorNotSupported;
  SkinnyArgs<E> a;
  a.K = g.K;
  a.kc = kc_elems;
  (void)laser_order;
  a.alpha = g.alpha; a.beta = g.beta;
  a.C = g.C;
  if (n_small) {  // X = A, Y = B
    a.L = g.M; a.S = (int32_t)g.N;
    a.X = g.A; a.sxl = g.rsA; a.sxk = g.csA;
    a.Y = g.B; a.syk = g.rsB; a.sys = g.csB;
    a.scl = g.rsC; a.scs = g.csC;
  } else {        // C^T = B^T A^T: X = B^T, Y = A^T
    a.L = g.N; a.S = (int32_t)g.M;
    a.X = g.B; a.sxl = g.csB; a.sxk = g.rsB;
    a.Y = g.A; a.syk = g.csA; a.sys = g.rsA;
    a.scl = g.csC; a.scs = g.rsC;
  }
  return a.sxk == 1 ? launch_s<E, true>(a, s) : launch_s<E, false>(a, s);
}
template hipError_t launch_gemm_skinny<float>(const GemmArgs<float> &, bool, int, hipStream_t);
template hipError_t launch_gemm_skinny<double>(const GemmArgs<double> &, bool, int, hipStream_t);
template hipError_t launch_gemm_skinny<int32_t>(const GemmArgs<int32_t> &, bool, int, hipStream_t);
template hipError_t launch_gemm_skinny<int64_t>(const GemmArgs<int64_t> &, bool, int, hipStream_t);

}  // namespace laser_hip
