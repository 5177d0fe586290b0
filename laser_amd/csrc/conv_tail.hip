// laser_amd/csrc/conv_tail.hip -- the pixel tail of the implicit-GEMM 3x3 convolution (round 5).
//
// The hand-scheduled convolution kernels (asmgen/f32_kernel.py conv=True) take whole 128-pixel tiles of every image; what is left --
// npix % 128 output pixels per image (C4: 64 of 3136) -- is 2 % of the work but, as a launch of its own, a LATENCY problem: few
// tiles, each a dependent chain over K = C_in * 9.  Round 3 ran it as images x kc-slices workgroups of the compiler-scheduled 64x64
// kernel into a workspace + an ordered combine pass (22.7 + 4.4 us at C4, two launches, 25 % of the matrix pipe busy:
// profiles/r05/rocprof_c4_v1/summary.md).  This kernel is built for the latency instead:
//   * one wave = one 32-channel x 32-pixel block over ONE kc = 512 slice (gemm.nim:150-158: a slice is an independent fused
//     chain from +0), v_mfma_f32_32x32x2_f32 with both operands loaded straight into the MFMA's lane layout -- no LDS staging, no
//     barrier in the K loop; lane (lo, hi) feeds A[m0 + lo][k + hi] (the filter, 16-byte loads of 4 consecutive k: L1 / L2 hits) and
//     B[k + hi][pixel lo] (one input element: the tap k = (c, kh, kw) of that pixel, zero in the padding -- conv2d_im2col.nim:62-87);
//     the loads of step s + 1 (32 k) are in flight while the matrix core works on step s;
//   * a workgroup = (image, 32-channel block), four waves sharing its pixel-block x slice tasks longest-first, so that C4's
//     32 x 8 workgroups are exactly one wave per SIMD with 640 of 1152 k each;
//   * the slices' sums meet in LDS and are folded IN ORDER -- run = S_0; run = run + S_1; ... -- by the wave that stores the block:
//     the same fused multiply-adds and the same unfused adds, in the same order, as the assembly kernels and as Laser
//     (gemm_ukernel_generic.nim:56-66 with alpha = 1, beta = 0): bit-identical.  No workspace, no combine launch.
// alpha = 1, beta = 0, no fused epilogue (what launch_conv_f32_asm takes); other tails keep the compiler-scheduled kernels.
#include "common.h"

namespace laser_hip {

namespace {

typedef __attribute__((ext_vector_type(16))) float ct_f32x16;
typedef __attribute__((ext_vector_type(4))) float ct_f32x4;

struct ConvTailArgs {
  const float *filt;   // [M][K], K = Cin * 9
  const float *img;    // [batch][Cin][H][W]
  float *out;          // [batch][M][npix]
  int64_t bsB, bsC;
  int32_t M, K, H, W, oW, npix, pH, pW, n_cut, nblk, nsl, kc;
};

constexpr int kCH = 16;          // MFMAs per step = 32 k
constexpr int kDepth = 4;        // register sets of the load ring

__global__ void __launch_bounds__(256) conv3x3_tail_kernel(const ConvTailArgs g) {
  extern __shared__ __attribute__((aligned(16))) float ct_lds[];      // [nsl][nblk][16][64]: slice sums of this workgroup's blocks
  const int t = threadIdx.x, lane = t & 63, lo = lane & 31, hi = lane >> 5, wave = t >> 6;
  const int mblks = (g.M + 31) / 32;
  const int b = (int)blockIdx.x / mblks, mb = (int)blockIdx.x - b * mblks;
  const float *img = g.img + (int64_t)b * g.bsB;
  const int m = mb * 32 + lo;
  const bool m_ok = m < g.M;
  const float *arow = g.filt + (int64_t)(m_ok ? m : g.M - 1) * g.K;
  const int HW = g.H * g.W;
  const int ntask = g.nblk * g.nsl;
  // tasks in the order (slice, block): every slice but the last is kc long, so the four waves' first tasks are the long ones and
  // the short last slices fill up behind them
  for (int task = wave; task < ntask; task += 4) {
    const int p = task / g.nblk, blk = task - p * g.nblk;
    const int k0 = p * g.kc, kend = min(g.K, k0 + g.kc);
    const int pix = g.n_cut + blk * 32 + lo;
    const bool p_ok = pix < g.npix;
    const int oh = p_ok ? pix / g.oW : 0, ow = p_ok ? pix - oh * g.oW : 0;
    const int ih0 = oh - g.pH, iw0 = ow - g.pW;
    const int org = ih0 * g.W + iw0;                   // the window origin of this lane's pixel, as an element index (may be negative)
    // this lane's running tap: k = kl, c = kl / 9, r = kl % 9 (advances by 2 per MFMA)
    int kl = k0 + hi, c = kl / 9, r = kl - c * 9;
    // Every load is UNCONDITIONAL (an invalid tap / piece reads element 0 of its operand instead and is replaced by zero where it is
    // consumed): a load under a lane condition becomes a branch round the load, and the compiler then waits for all outstanding loads
    // at every such branch -- the first build of this kernel ran one load at a time.  The validity bits travel with the ring.
    auto issue = [&](int ks, ct_f32x4 (&a)[8], float (&x)[kCH], unsigned &vmask) __attribute__((always_inline)) {
      unsigned vm = 0;
      // A: the 32 consecutive k of this step as eight 16-byte pieces (a piece beyond the slice's end is zero: kend % 4 == 0)
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int kq = ks + 4 * q;
        const bool ok = m_ok && kq < kend;
        a[q] = *reinterpret_cast<const ct_f32x4 *>(arow + (ok ? kq : 0));
        vm |= ok ? (1u << (16 + q)) : 0u;
      }
#pragma unroll
      for (int j = 0; j < kCH; j++) {
        const int kh = (r * 11) >> 5, kw = r - 3 * kh;
        const bool in = p_ok && kl < kend && (unsigned)(ih0 + kh) < (unsigned)g.H && (unsigned)(iw0 + kw) < (unsigned)g.W;
        x[j] = img[in ? org + c * HW + kh * g.W + kw : 0];
        vm |= in ? (1u << j) : 0u;
        kl += 2; r += 2;
        if (r >= 9) { r -= 9; c++; }
      }
      vmask = vm;
    };
    ct_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    auto compute = [&](const ct_f32x4 (&a)[8], const float (&x)[kCH], unsigned vm) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < kCH; j++) {
        // the element of the 32-k step this lane feeds MFMA j with: 2j + hi (piece (2j + hi) / 4; both halves' candidates are compile-time)
        const float araw = hi ? a[(2 * j + 1) >> 2][(2 * j + 1) & 3] : a[(2 * j) >> 2][(2 * j) & 3];
        const float av = (vm >> (16 + (j >> 1))) & 1u ? araw : 0.0f;
        const float xv = (vm >> j) & 1u ? x[j] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xv, acc, 0, 0, 0);
      }
    };
    const int nsteps = (kend - k0 + 2 * kCH - 1) / (2 * kCH);
    // A ring of kDepth register sets: the loads of steps s + 1 .. s + kDepth - 1 are in flight while step s multiplies.  With ONE
    // wave per SIMD (the whole point of this launch) nothing else hides the ~2 us a gather takes: two sets left every step waiting
    // (48 us for C4's tails, profiles/r05/conv_tail_ab_v1.jsonl); the wave has the SIMD's whole register file to itself.
    // (unconditional issues: a step past the end loads nothing -- every piece / element is masked by kend -- and is never multiplied)
    ct_f32x4 a[kDepth][8];
    float x[kDepth][kCH];
    unsigned vm[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth - 1; d++) issue(k0 + d * 2 * kCH, a[d], x[d], vm[d]);
#pragma unroll 1
    for (int s = 0; s < nsteps; s += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; d++) {
        // (no branch on s + d < nsteps: a step past the end is all-invalid -- it loads element 0 and multiplies zeros into the chain,
        // which leaves every accumulator as it is: x + 0 * 0 = x for every x the chain can hold)
        issue(k0 + (s + d + kDepth - 1) * 2 * kCH, a[(d + kDepth - 1) % kDepth], x[(d + kDepth - 1) % kDepth], vm[(d + kDepth - 1) % kDepth]);
        compute(a[d], x[d], vm[d]);
      }
    }
    float *dst = ct_lds + ((size_t)(p * g.nblk + blk) * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; i++) dst[i * 64] = acc[i];
  }
  __syncthreads();
  // ordered fold + store: wave w owns pixel blocks w, w + 4, ...
  float *out = g.out + (int64_t)b * g.bsC;
  for (int blk = wave; blk < g.nblk; blk += 4) {
    const int pix = g.n_cut + blk * 32 + lo;
    float run[16];
#pragma unroll
    for (int i = 0; i < 16; i++) run[i] = ct_lds[((size_t)blk * 16 + i) * 64 + lane];          // S_0: the first slice WRITES (beta == 0)
    for (int p = 1; p < g.nsl; p++) {
      const float *sp = ct_lds + ((size_t)(p * g.nblk + blk) * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 16; i++) run[i] = run[i] + sp[i * 64];                                // C += S_p, ascending p (gemm.nim:150-158)
    }
    if (pix < g.npix) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
        if (row < g.M) out[(int64_t)row * g.npix + pix] = run[i];
      }
    }
  }
}

}  // namespace

// Output pixels [a.col0, npix) of every image of a 3x3 / stride-1 convolution whose main part launch_conv_f32_asm took (GemmArgs as
// launch_conv_implicit_f32 builds them).  kc: the accumulation slice (512 in both modes: one-chain results have no order to keep).
// hipErrorNotSupported: not this kernel's class.
hipError_t launch_conv_tail_f32(const GemmArgs<float> &a, int kc, hipStream_t s) {
  if (a.ckH != 3 || a.ckW != 3 || a.csH != 1 || a.csW != 1 || a.cpH < 0 || a.cpW < 0) return hipErrorNotSupported;
  if (a.alpha != 1.0f || a.beta != 0.0f || a.bias != nullptr || a.act != 0 || a.cs_imgs != 0) return hipErrorNotSupported;
  const int64_t npix = a.N, ntail = npix - a.col0;
  if (ntail <= 0 || ntail > 128 || a.col0 < 0) return hipErrorNotSupported;
  if (a.csA != 1 || a.rsA != a.K || a.csC != 1 || a.rsC != npix || a.K % 36 != 0 || kc % 32 != 0 || kc < 32) return hipErrorNotSupported;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) != 0) return hipErrorNotSupported;      // (16-byte filter loads: K % 4 == 0 keeps every row aligned)
  const int64_t Cin = a.K / 9, nsl = (a.K + kc - 1) / kc, nblk = (ntail + 31) / 32, mblks = (a.M + 31) / 32;
  if ((double)Cin * a.cH * a.cW >= 2.0e9 || (double)a.M * npix >= 2.0e9 || npix >= ((int64_t)1 << 30)) return hipErrorNotSupported;
  const size_t lds = (size_t)nsl * nblk * 16 * 64 * sizeof(float);
  if (lds > ((size_t)64 << 10) || (int64_t)a.batch * mblks > 0x7fffffffLL) return hipErrorNotSupported;
  ConvTailArgs g;
  g.filt = a.A; g.img = a.B; g.out = a.C;
  g.bsB = a.bsB; g.bsC = a.bsC;
  g.M = (int32_t)a.M; g.K = (int32_t)a.K; g.H = a.cH; g.W = a.cW; g.oW = a.coW; g.npix = (int32_t)npix;
  g.pH = a.cpH; g.pW = a.cpW; g.n_cut = (int32_t)a.col0; g.nblk = (int32_t)nblk; g.nsl = (int32_t)nsl; g.kc = kc;
  hipLaunchKernelGGL(conv3x3_tail_kernel, dim3((unsigned)(a.batch * mblks)), dim3(256), lds, s, g);
  return hipGetLastError();
}

}  // namespace laser_hip
