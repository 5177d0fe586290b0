// laser_amd/csrc/gemm_small.hip -- the small-matrix path of gemm_strided (float32 / float64).
//
// The reference plans it ("Small matrix multiplication ... planned", README.md:257-263) and BASELINE configs[0] is the
// case: fp32 M = N = K = 128, 4 MFLOP -- launch- and latency-bound, not a tile problem.  The LDS-tiled kernels are the
// wrong tool there: a 64x64 tile leaves 4 workgroups for the whole chip, and their load -> LDS -> barrier -> MFMA
// pipeline exposes one memory round trip per K-tile.  Here every 32x32 (f64: 16x16) block of C is ONE WAVE on its own:
//   * no LDS, no barrier: the lane that feeds operand element (x, k) to the MFMA loads it straight from memory
//     (any strides: the "packing" of gemm_packing.nim:24-94 is the load's address arithmetic);
//   * all loads of a 64-k chunk are issued before its first MFMA, and the next chunk's loads are issued before the
//     current chunk's MFMAs: for K <= 128 the whole problem is in flight after ONE round trip to memory;
//   * the blocks spread over the chip (one wave per workgroup), the MFMA chain per block is the only serial part.
// Arithmetic: the same k-ascending fused-multiply-add chain per C element as the tiled kernels (an f32 / f64 MFMA is
// bitwise a k-ordered fma chain), restarted every kc and folded in slice order in laser-order mode => bit-identical
// to them and to the reference.  k beyond K is fed as zeros, like Laser's zero-padded panels (gemm_packing.nim:46-55).
// Also the batched-small engine: batch x (M, N <= 64) problems are batch x blocks independent waves.
// The operands may live in host memory mapped into the device (the host-pointer entry point's zero-copy staging
// buffer, capi.cpp): the kernel's loads then cross PCIe once, which replaces three blocking hipMemcpy calls.
#include "gemm_mfma_kernel.h"

namespace laser_hip {

template <typename E, bool EXACT>
__global__ void __launch_bounds__(64) gemm_small_kernel(const GemmArgs<E> g) {
  using M_ = Mma<E>;
  using Acc = typename M_::Acc;
  constexpr int MB = M_::MB, KS = M_::KS, ACC = M_::ACC;
  constexpr int KCH = 64;        // k per register chunk
  constexpr int NJ = KCH / KS;   // MFMA k-steps per chunk (f32: 32, f64: 16)
  const int lane = threadIdx.x, lo = M_::lx(lane), hi = M_::lk(lane);
  const int pid_n = blockIdx.x % g.tiles_n, pid_m = blockIdx.x / g.tiles_n;
  const int64_t bz = blockIdx.y;
  const int64_t m0 = (int64_t)pid_m * MB, n0 = (int64_t)pid_n * MB;
  // rows / columns beyond M / N only feed outputs that are never stored: clamp their addresses into the operand
  const int64_t row = min(m0 + lo, g.M - 1), col = min(n0 + lo, g.N - 1);
  const E *pa = g.A + bz * g.bsA + row * g.rsA;
  const E *pb = g.B + bz * g.bsB + col * g.csB;
  E *Cb = g.C + bz * g.bsC;
  const int64_t K = g.K;
  const E alpha = g.alpha, beta = g.beta;

  auto c_ptr = [&](int r, bool &ok) __attribute__((always_inline)) -> E * {
    const int64_t rr = m0 + M_::acc_row(r, lane), cc = n0 + M_::acc_col(lane);
    ok = rr < g.M && cc < g.N;
    return Cb + rr * g.rsC + cc * g.csC;
  };
  // beta*C0 exactly as the reference's epilogues: beta == 0 never reads C (gemm_ukernel_generic.nim:59-66, 107-115)
  auto scaled_c0 = [&](int r) __attribute__((always_inline)) -> E {
    if (beta == (E)0) return (E)0;
    bool ok;
    const E *p = c_ptr(r, ok);
    const E c0 = ok ? *p : (E)0;
    return beta == (E)1 ? c0 : M_::mul(c0, beta);
  };

  E fa[2][NJ], fb[2][NJ];
  // chunk c -> register set s: lane feeds k = k0 + KS*j + hi of k-step j; clamped address, zero beyond K
  auto load_chunk = [&](int64_t k0, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int64_t k = k0 + KS * j + hi;
      const int64_t kc_ = k < K ? k : K - 1;
      fa[s][j] = pa[kc_ * g.csA];
      fb[s][j] = pb[kc_ * g.rsB];
    }
  };
  Acc acc, run;
#pragma unroll
  for (int r = 0; r < ACC; r++) acc[r] = (E)0;
  if constexpr (EXACT) {
#pragma unroll
    for (int r = 0; r < ACC; r++) run[r] = scaled_c0(r);
  }
  const int nch = (int)((K + KCH - 1) / KCH);
  const int kc_chunks = EXACT ? g.kc / KCH : 0;
  load_chunk(0, 0);
  if (nch > 1) load_chunk(KCH, 1);
  auto mfma_chunk = [&](int64_t k0, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const bool in = k0 + KS * j + hi < K;  // (k-steps wholly beyond K add +0*0: the chain is unchanged)
      acc = M_::mma(in ? fa[s][j] : (E)0, in ? fb[s][j] : (E)0, acc);
    }
  };
  for (int c = 0; c < nch; c += 2) {
    if constexpr (EXACT) {
      // Laser's pc loop: the accumulator restarts at +0 every kc and the slice sum is added into C (gemm.nim:150-158)
      if (c > 0 && c % kc_chunks == 0) {
#pragma unroll
        for (int r = 0; r < ACC; r++) {
          run[r] = M_::add(run[r], M_::mul(alpha, acc[r]));
          acc[r] = (E)0;
        }
      }
    }
    mfma_chunk((int64_t)c * KCH, 0);
    if (c + 2 < nch) load_chunk((int64_t)(c + 2) * KCH, 0);
    if (c + 1 < nch) {
      if constexpr (EXACT) {
        if ((c + 1) % kc_chunks == 0) {
#pragma unroll
          for (int r = 0; r < ACC; r++) {
            run[r] = M_::add(run[r], M_::mul(alpha, acc[r]));
            acc[r] = (E)0;
          }
        }
      }
      mfma_chunk((int64_t)(c + 1) * KCH, 1);
      if (c + 3 < nch) load_chunk((int64_t)(c + 3) * KCH, 1);
    }
  }
  // epilogue: C = (beta*C0 or run) + alpha*acc, unfused (gemm_ukernel_generic.nim:68-76); optional fused bias / activation
#pragma unroll
  for (int r = 0; r < ACC; r++) {
    bool ok;
    E *p = c_ptr(r, ok);
    E base;
    if constexpr (EXACT)
      base = run[r];
    else
      base = scaled_c0(r);
    E out = M_::add(base, M_::mul(alpha, acc[r]));
    if (g.bias != nullptr || g.act != 0) {
      const int64_t rr = m0 + M_::acc_row(r, lane), cc = n0 + M_::acc_col(lane);
      if (g.bias != nullptr) out = M_::add(out, ok ? g.bias[bz * g.bsBias + rr * g.rsBias + cc * g.csBias] : (E)0);
      switch (g.act) {
        case 1: out = out > (E)0 ? out : (E)0; break;
        case 2: out = M_::tanh_(out); break;
        case 3: out = (E)1 / ((E)1 + M_::exp_(-out)); break;
        default: break;
      }
    }
    if (ok) *p = out;
  }
}

// Small-problem test: few enough 32x32 (16x16) blocks that one wave per block beats the tiled kernels -- at most one
// wave per CU for a single problem (256 blocks), any count for batches of tiny matrices (M, N <= 64: a 64x64 LDS tile
// would be mostly padding) -- and a K short enough that the per-block MFMA chain (K/2 x 64 cycles) stays in the
// microseconds.  hipErrorNotSupported: not small, use the tiled kernels.
int g_small_path = 1;  // knob (laser_hip_set_small_path)
bool gemm_small_takes(int elem_size, int64_t M, int64_t N, int64_t K, int64_t batch) {
  if (!g_small_path || (elem_size != 4 && elem_size != 8)) return false;
  const int mb = elem_size == 4 ? 32 : 16;
  const int64_t tm = (M + mb - 1) / mb, tn = (N + mb - 1) / mb;
  const bool tiny_batched = batch > 1 && M <= 64 && N <= 64;
  return (tm * tn * batch <= 256 || tiny_batched) && K <= 1024;
}
template <typename E>
hipError_t launch_gemm_small(const GemmArgs<E> &args, bool laser_order, int kc_elems, hipStream_t s) {
  using M_ = Mma<E>;
  if (args.M <= 0 || args.N <= 0 || args.K <= 0 || args.batch <= 0) return hipSuccess;
  if (!gemm_small_takes((int)sizeof(E), args.M, args.N, args.K, args.batch)) return hipErrorNotSupported;
  const int64_t tm = (args.M + M_::MB - 1) / M_::MB, tn = (args.N + M_::MB - 1) / M_::MB;
  GemmArgs<E> g = args;
  g.tiles_m = (int)tm;
  g.tiles_n = (int)tn;
  const bool exact = laser_order && args.K > kc_elems;
  g.kc = exact ? kc_elems : 0;
  if (std::is_same<E, float>::value) g_last_f32_cfg = -2;  // diagnostics: "the small-matrix kernel ran"
  dim3 grid((unsigned)(tm * tn), (unsigned)args.batch, 1), block(64, 1, 1);
  if (exact)
    hipLaunchKernelGGL((gemm_small_kernel<E, true>), grid, block, 0, s, g);
  else
    hipLaunchKernelGGL((gemm_small_kernel<E, false>), grid, block, 0, s, g);
  return hipGetLastError();
}
template hipError_t launch_gemm_small<float>(const GemmArgs<float> &, bool, int, hipStream_t);
template hipError_t launch_gemm_small<double>(const GemmArgs<double> &, bool, int, hipStream_t);

}  // namespace laser_hip
