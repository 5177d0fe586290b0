// laser_amd/csrc/gemm_valu.hip -- LDS-tiled VALU GEMM for the element types that have no matrix-core
// path with Laser's exact semantics yet: int32 / int64 (two's-complement wrap-around `c + a*b`,
// gemm_ukernel_avx512.nim:40-41,58-74, gemm_ukernel_avx2.nim:10-11) and float64 (FMA chain restarted
// every kc = 2048/8 = 256, gemm_tiling.nim:310).  Arbitrary strides on A, B, C are resolved in the
// global->LDS stage (the GPU analogue of pack_A/pack_B, gemm_packing.nim:24-94); ragged edges are
// zero-filled like the reference's zero-padded panels.  Integers are associative mod 2^n, so any
// summation order is bit-exact; float64 keeps the reference's slice order.
#include "common.h"

namespace laser_hip {

template <typename T> struct Arith;
template <> struct Arith<double> {
  static constexpr bool kSliced = true;
  static __device__ __forceinline__ double madd(double a, double b, double c) { return fma(a, b, c); }
  static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
  static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
};
template <> struct Arith<float> {
  static constexpr bool kSliced = true;
  static __device__ __forceinline__ float madd(float a, float b, float c) { return fmaf(a, b, c); }
  static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
  static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
};
template <> struct Arith<int32_t> {
  static constexpr bool kSliced = false;
  static __device__ __forceinline__ int32_t madd(int32_t a, int32_t b, int32_t c) {
    return (int32_t)((uint32_t)c + (uint32_t)a * (uint32_t)b);
  }
  static __device__ __forceinline__ int32_t mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
  static __device__ __forceinline__ int32_t add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
};
template <> struct Arith<int64_t> {
  static constexpr bool kSliced = false;
  static __device__ __forceinline__ int64_t madd(int64_t a, int64_t b, int64_t c) {
    return (int64_t)((uint64_t)c + (uint64_t)a * (uint64_t)b);
  }
  static __device__ __forceinline__ int64_t mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
  static __device__ __forceinline__ int64_t add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
};

constexpr int VT = 64;   // C tile edge
constexpr int VK = 16;   // k per LDS stage
constexpr int VTHREADS = 256;

template <typename T>
__global__ void __launch_bounds__(VTHREADS) gemm_valu_kernel(const GemmArgs<T> g) {
  using AR = Arith<T>;
  __shared__ T sA[2][VK][VT + 1];
  __shared__ T sB[2][VK][VT + 1];
  const int t = threadIdx.x, tx = t % 16, ty = t / 16;
  const int64_t m0 = (int64_t)blockIdx.y * VT, n0 = (int64_t)blockIdx.x * VT;
  const int64_t bz = blockIdx.z;
  const T *A = g.A + bz * g.bsA;
  const T *B = g.B + bz * g.bsB;
  T *C = g.C + bz * g.bsC;
  // lanes run along whichever axis has the smaller stride (coalescing)
  const bool a_along_k = (g.csA < 0 ? -g.csA : g.csA) <= (g.rsA < 0 ? -g.rsA : g.rsA);
  const bool b_along_k = (g.rsB < 0 ? -g.rsB : g.rsB) < (g.csB < 0 ? -g.csB : g.csB);

  T acc[4][4], run[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (T)0;

  auto cptr = [&](int i, int j, bool &ok) -> T * {
    const int64_t r = m0 + ty * 4 + i, c = n0 + tx * 4 + j;
    ok = r < g.M && c < g.N;
    return C + r * g.rsC + c * g.csC;
  };
  // beta*C0 with the reference's case split (never read C when beta == 0)
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      T v = (T)0;
      if (g.beta != (T)0) {
        bool ok;
        const T *p = cptr(i, j, ok);
        const T c0 = ok ? *p : (T)0;
        v = (g.beta == (T)1) ? c0 : AR::mul(c0, g.beta);
      }
      run[i][j] = v;
    }
  auto fold = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        run[i][j] = AR::add(run[i][j], g.alpha == (T)1 ? acc[i][j] : AR::mul(g.alpha, acc[i][j]));
        acc[i][j] = (T)0;
      }
  };

  T ra[4], rb[4];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = t + i * VTHREADS;
      const int ka = a_along_k ? idx % VK : idx / VT, ma = a_along_k ? idx / VK : idx % VT;
      const int64_t r = m0 + ma, k = k0 + ka;
      ra[i] = (r < g.M && k < g.K) ? A[r * g.rsA + k * g.csA] : (T)0;
      const int kb = b_along_k ? idx % VK : idx / VT, nb = b_along_k ? idx / VK : idx % VT;
      const int64_t c = n0 + nb, k2 = k0 + kb;
      rb[i] = (c < g.N && k2 < g.K) ? B[k2 * g.rsB + c * g.csB] : (T)0;
    }
  };
  auto sstore = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = t + i * VTHREADS;
      const int ka = a_along_k ? idx % VK : idx / VT, ma = a_along_k ? idx / VK : idx % VT;
      sA[st][ka][ma] = ra[i];
      const int kb = b_along_k ? idx % VK : idx / VT, nb = b_along_k ? idx / VK : idx % VT;
      sB[st][kb][nb] = rb[i];
    }
  };

  const int nkt = (int)((g.K + VK - 1) / VK);
  const int kc_tiles = (AR::kSliced && g.kc > 0) ? g.kc / VK : 0;
  int until_fold = kc_tiles;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int st = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) gload((int64_t)(kt + 1) * VK);
#pragma unroll
    for (int k = 0; k < VK; k++) {
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = sA[st][k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = sB[st][k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = AR::madd(a[i], b[j], acc[i][j]);
    }
    if (AR::kSliced && --until_fold == 0 && more) {
      until_fold = kc_tiles;
      fold();
    }
    if (more) sstore(st ^ 1);
    __syncthreads();
  }
  fold();
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      bool ok;
      T *p = cptr(i, j, ok);
      if (ok) *p = run[i][j];
    }
}

template <typename T>
hipError_t launch_gemm_valu(const GemmArgs<T> &args, bool laser_order, hipStream_t s) {
  if (args.M <= 0 || args.N <= 0 || args.K <= 0 || args.batch <= 0) return hipSuccess;
  GemmArgs<T> a = args;
  a.kc = laser_order ? (int)(2048 / sizeof(T)) : 0;  // gemm_tiling.nim:310
  dim3 grid((unsigned)((a.N + VT - 1) / VT), (unsigned)((a.M + VT - 1) / VT), (unsigned)a.batch);
  hipLaunchKernelGGL(gemm_valu_kernel<T>, grid, dim3(VTHREADS), 0, s, a);
  return hipGetLastError();
}

template hipError_t launch_gemm_valu<double>(const GemmArgs<double> &, bool, hipStream_t);
template hipError_t launch_gemm_valu<float>(const GemmArgs<float> &, bool, hipStream_t);
template hipError_t launch_gemm_valu<int32_t>(const GemmArgs<int32_t> &, bool, hipStream_t);
template hipError_t launch_gemm_valu<int64_t>(const GemmArgs<int64_t> &, bool, hipStream_t);

}  // namespace laser_hip
