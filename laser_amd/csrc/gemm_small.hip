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

// what a lane loads for k >= K: zeros, like Laser's zero-padded panels (gemm_packing.nim:46-55) -- by ADDRESS (the
// load is pointed here), so no select sits between a load and the MFMA that consumes it
__device__ __attribute__((aligned(16))) const double g_small_zero[2] = {0.0, 0.0};

// AV / BV: the operand is unit-stride along k (A row-major: colStrideA == 1; B passed transposed: rowStrideB == 1), so
// a lane fetches 16 bytes = 4 (f64: 2) consecutive k with one load and keeps the ones its MFMA half consumes (the lane
// halves / quarters take alternate k: half of a vector is used, but the instruction count halves).  Otherwise one
// scalar load per (lane, k-step).
template <typename E, bool EXACT, bool AV, bool BV>
__global__ void __launch_bounds__(64) gemm_small_kernel(const GemmArgs<E> g) {
  using M_ = Mma<E>;
  using Acc = typename M_::Acc;
  using Vec = typename M_::Vec;
  constexpr int MB = M_::MB, KS = M_::KS, ACC = M_::ACC, EPV = M_::EPV;
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

  // register image of a chunk: element [j] = what this lane feeds to k-step j (k = k0 + KS*j + hi).
  // ONE code path for full and ragged chunks (a 5 us kernel cannot afford several unrolled copies of itself in the
  // instruction cache): the address is one v_mad_i64_i32 off the row / column base (byte strides fit 32 bits: the
  // launcher checks), or the address of a zero constant for k >= K.
  E fa[2][NJ], fb[2][NJ];
  const int Km1 = (int)K - 1;
  const int sab = (int)(g.csA * (int64_t)sizeof(E)), sbb = (int)(g.rsB * (int64_t)sizeof(E));  // byte strides along k
  auto load_operand = [&](E (&f)[NJ], const E *p, int skb, bool vec, int k0) __attribute__((always_inline)) {
    // (explicit global address space: a select between two generic pointers would lower to flat loads)
    typedef const __attribute__((address_space(1))) char *gptr;
    typedef const __attribute__((address_space(1))) Vec *gvec;
    typedef const __attribute__((address_space(1))) E *gelem;
    const gptr pc = (gptr)(uintptr_t)p;
    const gptr zero = (gptr)(uintptr_t)g_small_zero;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if (vec) {
        // unit stride along k: the 16-byte piece holding k-offset KS*j + hi (K % EPV == 0: the launcher checks).
        // f32 (KS = 2, EPV = 4): steps 2q, 2q+1 share piece q (one load after CSE), element 2(j%2) + hi;
        // f64 (KS = 4, EPV = 2): lane quarters hi = 0,1 / 2,3 take pieces 2j / 2j + 1, element hi % 2.
        const int kp = sizeof(E) == 4 ? k0 + EPV * (j / 2) : k0 + KS * j + 2 * (hi >> 1);
        const gptr ad = pc + (int64_t)kp * (int64_t)sizeof(E);
        const Vec v = *(gvec)(kp <= Km1 ? ad : zero);
        if constexpr (sizeof(E) == 4)
          f[j] = hi ? v[2 * (j & 1) + 1] : v[2 * (j & 1)];
        else
          f[j] = (hi & 1) ? v[1] : v[0];
      } else {
        const int k = k0 + KS * j + hi;
        const gptr ad = pc + (int64_t)k * (int64_t)skb;
        f[j] = *(gelem)(k <= Km1 ? ad : zero);
      }
    }
  };
  auto load_chunk = [&](int c, int s) __attribute__((always_inline)) {
    load_operand(fa[s], pa, sab, AV, c * KCH);
    load_operand(fb[s], pb, sbb, BV, c * KCH);
  };
  Acc acc, run;
#pragma unroll
  for (int r = 0; r < ACC; r++) acc[r] = (E)0;
  if constexpr (EXACT) {
#pragma unroll
    for (int r = 0; r < ACC; r++) run[r] = scaled_c0(r);
  }
  const int nch = (int)((K + KCH - 1) / KCH);
  const int kc_chunks = EXACT ? g.kc / KCH : 1;
  auto mfma_chunk = [&](int c, int s) __attribute__((always_inline)) {
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
#pragma unroll
    for (int j = 0; j < NJ; j++) acc = M_::mma(fa[s][j], fb[s][j], acc);  // (k >= K: +0 * +0, the chain is unchanged)
  };
  load_chunk(0, 0);
  for (int c = 0; c < nch; c += 2) {
    if (c + 1 < nch) load_chunk(c + 1, 1);
    mfma_chunk(c, 0);
    if (c + 2 < nch) load_chunk(c + 2, 0);
    if (c + 1 < nch) mfma_chunk(c + 1, 1);
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
  // host-mapped run (capi.cpp: gemm_host): tell the polling host this block is done -- release at system scope first, so
  // the block's stores to C have left the device before its flag can be seen
  if (g.done_flags != nullptr) {
    __threadfence_system();
    if (lane == 0)
      __hip_atomic_store(g.done_flags + (bz * gridDim.x + blockIdx.x), g.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Dispatch rule, from measurements with a compiled caller (tests/cpp/small_sweep.cpp, small_gemm_bench.cpp;
// profiles/r02/small_sweep_v1.jsonl, small_gemm_v13.jsonl).  The kernel's serial part is the per-block MFMA chain
// (K/2 x 64 cycles) and its loads are not shared between blocks, so it pays only where the tiled kernels cannot amortise
// anything:
//   host-mapped operands (the zero-copy staging of the host-pointer entry point, `mapped`), <= 256 blocks, K <= 1024:
//     every load crosses PCIe, and this kernel has them all in flight after one round trip where the tiled kernel pays
//     one round trip per K-tile (128^3 end to end: 22.9 vs 52 us);
//   device-resident BATCHES of matrices up to 64x64, K <= 128: batch x blocks independent waves (1000 x 32^3: 6.4 vs
//     7.2 us per launch).
//   A single device-resident problem never: on all 18 shapes of the sweep (32^3 .. 512x512x128) the 64x64-tile kernel,
//     whose 4 waves share every operand element through LDS, is 4-30 % faster (128^3: 5.9 vs 6.7 us per launch).
// hipErrorNotSupported: not this kernel's case, use the tiled kernels.
std::atomic<int> g_small_path{1};  // knob (laser_hip_set_small_path)
bool gemm_small_takes(int elem_size, int64_t M, int64_t N, int64_t K, int64_t batch, bool mapped) {
  if (!g_small_path || (elem_size != 4 && elem_size != 8)) return false;
  const int mb = elem_size == 4 ? 32 : 16;
  const int64_t tm = (M + mb - 1) / mb, tn = (N + mb - 1) / mb;
  if (mapped) return tm * tn * batch <= 256 && K <= 1024;
  return batch > 1 && M <= 64 && N <= 64 && K <= 128;
}
template <typename E>
hipError_t launch_gemm_small(const GemmArgs<E> &args, bool laser_order, int kc_elems, hipStream_t s, bool mapped) {
  using M_ = Mma<E>;
  if (args.M <= 0 || args.N <= 0 || args.K <= 0 || args.batch <= 0) return hipSuccess;
  if (!gemm_small_takes((int)sizeof(E), args.M, args.N, args.K, args.batch, mapped)) return hipErrorNotSupported;
  const int64_t tm = (args.M + M_::MB - 1) / M_::MB, tn = (args.N + M_::MB - 1) / M_::MB;
  GemmArgs<E> g = args;
  g.tiles_m = (int)tm;
  g.tiles_n = (int)tn;
  const bool exact = laser_order && args.K > kc_elems;
  g.kc = exact ? kc_elems : 0;
  if (std::is_same<E, float>::value) g_last_f32_cfg = -2;  // diagnostics: "the small-matrix kernel ran"
  // 16-byte loads along k need an aligned base and strides that keep every row / column on a 16-byte boundary
  constexpr int64_t EPV = 16 / sizeof(E);
  auto aligned = [&](const E *p, int64_t sx, int64_t bs) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && sx % EPV == 0 && bs % EPV == 0; };
  const bool av = args.csA == 1 && args.K % EPV == 0 && aligned(args.A, args.rsA, args.bsA);
  const bool bv = args.rsB == 1 && args.K % EPV == 0 && aligned(args.B, args.csB, args.bsB);
  // 32-bit byte strides along k (one v_mad_i64_i32 per address); anything wider goes to the tiled kernels
  auto fits = [](int64_t st) { return st * (int64_t)sizeof(E) < (1ll << 31) && st * (int64_t)sizeof(E) > -(1ll << 31); };
  if (!fits(args.csA) || !fits(args.rsB) || args.K >= (1ll << 30)) return hipErrorNotSupported;
  dim3 grid((unsigned)(tm * tn), (unsigned)args.batch, 1), block(64, 1, 1);
#define LH_SMALL(EX, AVV, BVV) hipLaunchKernelGGL((gemm_small_kernel<E, EX, AVV, BVV>), grid, block, 0, s, g)
  if (exact) {
    if (av && bv) LH_SMALL(true, true, true);
    else if (av) LH_SMALL(true, true, false);
    else if (bv) LH_SMALL(true, false, true);
    else LH_SMALL(true, false, false);
  } else {
    if (av && bv) LH_SMALL(false, true, true);
    else if (av) LH_SMALL(false, true, false);
    else if (bv) LH_SMALL(false, false, true);
    else LH_SMALL(false, false, false);
  }
#undef LH_SMALL
  return hipGetLastError();
}
template hipError_t launch_gemm_small<float>(const GemmArgs<float> &, bool, int, hipStream_t, bool);
template hipError_t launch_gemm_small<double>(const GemmArgs<double> &, bool, int, hipStream_t, bool);

}  // namespace laser_hip
