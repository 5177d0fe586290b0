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
  int32_t M, K, H, W, oW, npix, pH, pW, n_cut, nblk, nsl, kc, ktab;
};

constexpr int kCH = 16;          // MFMAs per step = 32 k
constexpr int kDepth = 4;        // register sets of the load ring
constexpr int kAStep = 32 * 33;  // floats of one staged filter step in LDS

// a wave-uniform pointer as a bounds-checked raw buffer (what lies beyond `bytes` reads as 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ct_rsrc(const void *p, int64_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(p);
  const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const int nrec = __builtin_amdgcn_readfirstlane((int)(uint32_t)(bytes > 0x7fffffffll ? 0x7fffffffll : (bytes < 0 ? 0 : bytes)));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bu), 0, nrec, 0x00020000);
}

// The K loop runs with ONE wave per SIMD, so nothing hides what the wave itself does beside its MFMAs: a vector instruction beside
// the MFMA stream costs matrix-pipe time.  Per MFMA this kernel spends four vector instructions and one LDS read:
//   * the tap (c, kh, kw) of every k is a table in LDS, built once per workgroup: {element offset (c*H + kh)*W + kw, r = kh*3 + kw}
//     (r = 9 beyond K); a lane reads the entry of ITS k (k + hi) with one ds_read_b64;
//   * whether that tap exists for the lane's pixel is bit r of a per-lane mask (padding / beyond the image / no pixel): v_bfe_i32 gives
//     0 or -1, OR-ed into the offset -- an offset of -4 is out of the buffer's range and reads as 0, no select, no branch;
//   * the filter block of a step is fetched coalesced and passes through a per-wave LDS buffer into the MFMA lane layout (below);
// Loads are never under a lane condition (a branch round a load makes the compiler drain every outstanding load there: the first
// build of this kernel ran one load at a time) and kDepth - 1 steps are in flight while one multiplies.
__global__ void __launch_bounds__(256) conv3x3_tail_kernel(const ConvTailArgs g) {
  extern __shared__ __attribute__((aligned(16))) float ct_lds[];      // [nsl][nblk][16][64] slice sums, then the tap table int2[ktab]
  const int t = threadIdx.x, lane = t & 63, lo = lane & 31, hi = lane >> 5, wave = t >> 6;
  const int mblks = (g.M + 31) / 32;
  const int b = (int)blockIdx.x / mblks, mb = (int)blockIdx.x - b * mblks;
  const int HW = g.H * g.W;
  int2 *tab = reinterpret_cast<int2 *>(ct_lds + (size_t)g.nsl * g.nblk * 16 * 64);
  // this wave's filter-step buffer [32 k][33]: element (k, row) at k * 33 + row (the pitch puts the 32 rows of one k, and the k of
  // one row, into different banks).  ONE buffer: a step's fragments are read back into registers right behind their stores, and a
  // wave's LDS operations execute in order, so the next step's stores cannot overtake those reads
  float *abuf = ct_lds + (size_t)g.nsl * g.nblk * 16 * 64 + (size_t)g.ktab * 2 + (size_t)wave * kAStep;
  for (int k = t; k < g.ktab; k += 256) {
    const int c = k / 9, r = k - c * 9, kh = (r * 11) >> 5, kw = r - 3 * kh;
    tab[k] = k < g.K ? make_int2(c * HW + kh * g.W + kw, r) : make_int2(0, 9);
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsA = ct_rsrc(g.filt, (int64_t)g.M * g.K * 4);
  const __amdgpu_buffer_rsrc_t rsB = ct_rsrc(g.img + (int64_t)b * g.bsB, (int64_t)(g.K / 9) * HW * 4);
  // The filter block of a step -- 32 rows x 32 k -- is fetched COALESCED (eight lanes per row: 128 contiguous bytes, 8 rows per
  // instruction, four instructions) and passes through this wave's LDS buffer into the MFMA lane layout.  Loading the fragments
  // straight from global memory (a lane per row: 32 cache lines per instruction, eight instructions a step) was what the first builds
  // spent their time on: the kernel's time followed the cache-line look-ups per step, not the MFMAs or the VALU
  // (profiles/r05/conv_tail_study_v1.md).  Rows beyond M and k beyond K start out of the buffer's range (read as 0).
  int arow[4], awr[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (lane >> 3) + 8 * i, kq = (lane & 7) * 4, mrow = mb * 32 + row;
    arow[i] = mrow < g.M ? (mrow * g.K + kq) * 4 : (int)0x80000000;
    awr[i] = kq * 33 + row;                    // LDS float index of element (kq, row); (kq + e, row) is e * 33 further
  }
  const int ard = hi * 33 + lo;                // LDS float index of element (hi, lo); MFMA j reads (2j + hi, lo) = ard + 66 j
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
    unsigned inv = 1u << 9;                            // bit r: tap r does not exist for this pixel (bit 9: k beyond K)
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const int kh = r / 3, kw = r - 3 * kh;
      const bool in = p_ok && (unsigned)(ih0 + kh) < (unsigned)g.H && (unsigned)(iw0 + kw) < (unsigned)g.W;
      inv |= in ? 0u : (1u << r);
    }
    // a step's loads are issued in two phases one step apart: its table entries are READ from LDS while the step before it multiplies
    // (a wave alone on its SIMD has nothing else to cover an LDS read's ~100 cycles: sixteen reads each waited for where its
    // address arithmetic starts were a third of the first build's time), the addresses and the loads follow from registers
    auto tabread = [&](int ks, int2 (&e)[kCH]) __attribute__((always_inline)) {
      const int2 *tk = tab + ks + hi;
#pragma unroll
      for (int j = 0; j < kCH; j++) e[j] = tk[2 * j];
    };
    auto issue = [&](int ks, const int2 (&e)[kCH], ct_f32x4 (&a)[4], float (&x)[kCH]) __attribute__((always_inline)) {
      const bool kin = ks + (lane & 7) * 4 < kend;                     // (kend % 4 == 0: a 16-byte piece is inside the slice or beyond it)
#pragma unroll
      for (int i = 0; i < 4; i++)
        a[i] = __builtin_bit_cast(ct_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? arow[i] + ks * 4 : (int)0x80000000, 0, 0));
#pragma unroll
      for (int j = 0; j < kCH; j++) {
        const int bad = __builtin_amdgcn_sbfe(inv, e[j].y, 1);         // 0 (the tap exists) or -1
        x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, ((org + e[j].x) | bad) << 2, 0, 0));
      }
    };
    ct_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    // stage: the step's four pieces -> this wave's LDS buffer (k-major), then its sixteen fragment elements back into registers.
    // Only this wave touches the buffer and a wave's LDS operations execute in order: no barrier.
    auto stage = [&](const ct_f32x4 (&a)[4], float (&af)[kCH]) __attribute__((always_inline)) {
      float *ab = abuf;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int el = 0; el < 4; el++) ab[awr[i] + 33 * el] = a[i][el];
#pragma unroll
      for (int j = 0; j < kCH; j++) af[j] = ab[ard + 66 * j];
    };
    auto compute = [&](const float (&af)[kCH], const float (&x)[kCH]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < kCH; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], x[j], acc, 0, 0, 0);
    };
    // steps of 32 k; a step that starts at or beyond kend is issued (the ring is unconditional; its loads hit valid table entries and
    // in-range or zero-reading offsets) but never multiplied: the guard below is wave-uniform and wraps no load
    const int nsteps = (kend - k0 + 2 * kCH - 1) / (2 * kCH);
    ct_f32x4 a[kDepth][4];
    float x[kDepth][kCH];
    float af[2][kCH];
    // (the scheduling barriers pin the ISSUE ORDER of the steps: the vector-memory counter retires in order, so a step can be waited
    // for with the later ones still in flight only if its loads really were issued first -- left alone the compiler sorted the
    // prologue's loads its own way and the loop waited for vmcnt(0))
    int2 e[2][kCH];
    tabread(k0, e[0]);
#pragma unroll
    for (int d = 0; d < kDepth - 1; d++) {
      tabread(k0 + (d + 1) * 2 * kCH, e[(d + 1) & 1]);
      issue(k0 + d * 2 * kCH, e[d & 1], a[d], x[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage(a[0], af[0]);
    __builtin_amdgcn_sched_barrier(0);
    static_assert(kDepth % 2 == 0, "the table-entry sets and the fragment sets alternate with the step's parity");
#pragma unroll 1
    for (int s = 0; s < nsteps; s += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; d++) {
        // step n = s + d multiplies; step n + kDepth - 1 is issued from the table entries read one step ago, the entries of step
        // n + kDepth are read, and the filter block of step n + 1 (loaded kDepth - 2 steps ago) passes through LDS into the fragment
        // registers the next step multiplies from
        issue(k0 + (s + d + kDepth - 1) * 2 * kCH, e[(d + kDepth - 1) & 1], a[(d + kDepth - 1) % kDepth], x[(d + kDepth - 1) % kDepth]);
        __builtin_amdgcn_sched_barrier(0);
        tabread(k0 + (s + d + kDepth) * 2 * kCH, e[(d + kDepth) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        stage(a[(d + 1) % kDepth], af[(d + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + d < nsteps) compute(af[d & 1], x[d]);
        __builtin_amdgcn_sched_barrier(0);
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
  const int64_t Cin = a.K / 9, nsl = (a.K + kc - 1) / kc, nblk = (ntail + 31) / 32, mblks = (a.M + 31) / 32;
  if ((double)Cin * a.cH * a.cW >= 2.0e9 || (double)a.M * npix >= 2.0e9 || npix >= ((int64_t)1 << 30)) return hipErrorNotSupported;
  // the tap table covers every k a step of the ring can name: the slices + the steps issued past the last one
  // (the ring issues up to kDepth - 1 + kDepth steps beyond a slice's last one, each naming 2 * kCH entries: covered for every slice length)
  const int64_t ktab = nsl * kc + (int64_t)(2 * kDepth + 1) * 2 * kCH;
  const size_t lds = (size_t)nsl * nblk * 16 * 64 * sizeof(float) + (size_t)ktab * sizeof(int2) + (size_t)4 * kAStep * sizeof(float);
  if (lds > ((size_t)64 << 10) || (int64_t)a.batch * mblks > 0x7fffffffLL) return hipErrorNotSupported;      // (longer reductions: the round-3 tail forms)
  if ((double)a.M * a.K * 4.0 >= 2147483648.0 || (double)Cin * a.cH * a.cW * 4.0 >= 2147483648.0) return hipErrorNotSupported;   // 31-bit byte offsets
  ConvTailArgs g;
  g.filt = a.A; g.img = a.B; g.out = a.C;
  g.bsB = a.bsB; g.bsC = a.bsC;
  g.M = (int32_t)a.M; g.K = (int32_t)a.K; g.H = a.cH; g.W = a.cW; g.oW = a.coW; g.npix = (int32_t)npix;
  g.pH = a.cpH; g.pW = a.cpW; g.n_cut = (int32_t)a.col0; g.nblk = (int32_t)nblk; g.nsl = (int32_t)nsl; g.kc = kc; g.ktab = (int32_t)ktab;
  hipLaunchKernelGGL(conv3x3_tail_kernel, dim3((unsigned)(a.batch * mblks)), dim3(256), lds, s, g);
  return hipGetLastError();
}

}  // namespace laser_hip
