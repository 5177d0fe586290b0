// laser_amd/csrc/conv_tail.hip -- the pixel tail of the implicit-GEMM 3x3 convolution (round 5).
//
// The hand-scheduled convolution kernels (asmgen/f32_kernel.py conv=True) take whole 128-pixel tiles of every image; what is left --
// npix % 128 output pixels per image (C4: 64 of 3136) -- is 2 % of the work but, as a launch of its own, a LATENCY problem: few
// tiles, each a dependent chain over K = C_in * 9.  Round 3 ran it as images x kc-slices workgroups of the compiler-scheduled 64x64
// kernel into a workspace + an ordered combine pass (22.7 + 4.4 us at C4, two launches, 25 % of the matrix pipe busy:
// profiles/r05/rocprof_c4_v1/summary.md).  This kernel is built for the latency instead:
//   * one wave = one 32-channel x 32-pixel block over ONE kc = 512 slice (gemm.nim:150-158: a slice is an independent fused
//     chain from +0) on v_mfma_f32_32x32x2_f32: lane (lo, hi) feeds A[m0 + lo][k + hi] (the filter) and B[k + hi][pixel lo] (one
//     input element: the tap k = (c, kh, kw) of that pixel, zero in the padding -- conv2d_im2col.nim:62-87); no barrier in the K loop,
//     the loads of three steps (32 k each) in flight while the matrix core works on one;
//   * a workgroup = (image, 32-channel block) with one wave per pixel-block x slice task (four to eight waves; more tasks than
//     that are shared longest-first): at C4 32 x 8 workgroups of six waves -- every SIMD of the chip has one 512-k chain;
//   * the slices' sums meet in LDS and are folded IN ORDER -- run = S_0; run = run + S_1; ... -- by the waves that store the block:
//     the same fused multiply-adds and the same unfused adds, in the same order, as the assembly kernels and as Laser
//     (gemm_ukernel_generic.nim:56-66 with alpha = 1, beta = 0): bit-identical.  No workspace, no combine launch.
// What the K loop is built around (measured with stamps and ablations of this kernel, profiles/r05/conv_tail_study_v1.md):
//   1. a fragment load with a lane per filter row touches 32 cache lines per instruction -- the filter block of a step is fetched
//      COALESCED and passes through a per-wave LDS buffer into the MFMA lane layout (28 -> 22 us);
//   2. a step's sixteen MFMAs are a dependent chain and a wave issues in order -- the step's side work stands BEHIND EACH MFMA,
//      a few instructions at a time, not in phases between the steps (2 090 -> 1 970 cycles a step);
//   3. a vector-ALU instruction beside the MFMA stream costs matrix-pipe time (here: four per gather = 800 of a step's
//      1 970 cycles against the MFMAs' 1 024) -- the loop of the common case has NO address arithmetic on the vector unit: the
//      gather offsets of a lane repeat every 18 k (two channels), so nine registers hold them for the whole slice, the channel
//      pair advances in the instruction's SCALAR offset, and padding is an offset beyond the buffer's range (reads as 0).
// alpha = 1, beta = 0, no fused epilogue (what launch_conv_f32_asm takes); other tails keep the compiler-scheduled kernels.
#include "common.h"
#include <algorithm>

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
#ifdef CT_TIMING
  unsigned long long *dbg;   // probe build only (scripts/probes/conv_tail_timing.hip): [workgroup][8 waves][8] s_memtime stamps
#endif
};

#ifdef CT_TIMING
#define CT_STAMP(i) do { if (lane == 0) g.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CT_STAMP(i) do { } while (0)
#endif

constexpr int kCH = 16;          // MFMAs per step = 32 k
constexpr int kDepth = 4;        // register sets of the load ring
constexpr int kAStep = 32 * 33;  // floats of one staged filter step in LDS

// a wave-uniform pointer as a bounds-checked raw buffer (an OFFSET REGISTER value beyond `bytes` reads as 0; the instruction's
// scalar offset is not part of that check)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ct_rsrc(const void *p, int64_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(p);
  const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const int nrec = __builtin_amdgcn_readfirstlane((int)(uint32_t)(bytes > 0x7fffffffll ? 0x7fffffffll : (bytes < 0 ? 0 : bytes)));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bu), 0, nrec, 0x00020000);
}

// kFast: every slice is a whole number of 32-k steps (C4: 512, 512, 128) -- the loop without vector address arithmetic.
// !kFast: ragged slices -- the tap (c, kh, kw) of every k is a table in LDS ({element offset, r = kh*3 + kw}, r = 9 beyond K), a
// lane reads the entry of ITS k one step ahead and forms the offset with four vector instructions per gather; a filter piece beyond
// the slice's end reads as 0 through a per-lane select.
// Loads are never under a lane condition (a branch round a load makes the compiler drain every outstanding load there: the first
// build of this kernel ran one load at a time), and there is no exit out of the unrolled round either: a join inside it makes the
// compiler's wait-count pass wait for every MFMA with one step in flight instead of three.
template <bool kFast>
__global__ void __launch_bounds__(512) conv3x3_tail_kernel(const ConvTailArgs g) {
  extern __shared__ __attribute__((aligned(16))) float ct_lds[];      // [nsl][nblk][16][64] slice sums, the tap table int2[ktab], the waves' filter buffers
  const int t = threadIdx.x, lane = t & 63, lo = lane & 31, hi = lane >> 5, nwave = (int)blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // (a scalar: what follows from it -- task, slice, step counts -- lives on the scalar unit)
  const int mblks = (g.M + 31) / 32;
  const int b = (int)blockIdx.x / mblks, mb = (int)blockIdx.x - b * mblks;
  const int HW = g.H * g.W;
  CT_STAMP(0);
  int2 *tab = reinterpret_cast<int2 *>(ct_lds + (size_t)g.nsl * g.nblk * 16 * 64);
  // this wave's filter-step buffer [32 k][33]: element (k, row) at k * 33 + row (the pitch puts the 32 rows of one k, and the k of
  // one row, into different banks).  ONE buffer: a step's fragments are read back into registers behind their stores, and a wave's
  // LDS operations execute in order, so the next step's stores cannot overtake those reads
  float *abuf = ct_lds + (size_t)g.nsl * g.nblk * 16 * 64 + (size_t)g.ktab * 2 + (size_t)wave * kAStep;
  if constexpr (!kFast) {
    for (int k = t; k < g.ktab; k += (int)blockDim.x) {
      const int c = k / 9, r = k - c * 9, kh = (r * 11) >> 5, kw = r - 3 * kh;
      tab[k] = k < g.K ? make_int2(c * HW + kh * g.W + kw, r) : make_int2(0, 9);
    }
    __syncthreads();
  }
  CT_STAMP(1);
  const __amdgpu_buffer_rsrc_t rsA = ct_rsrc(g.filt, (int64_t)g.M * g.K * 4);
  const __amdgpu_buffer_rsrc_t rsB = ct_rsrc(g.img + (int64_t)b * g.bsB, (int64_t)(g.K / 9) * HW * 4);
  // The filter block of a step -- 32 rows x 32 k -- is fetched coalesced: eight lanes per row (128 contiguous bytes), 8 rows per
  // instruction, four instructions.  Rows beyond M start out of the buffer's range (read as 0).
  int arow[4], awr[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (lane >> 3) + 8 * i, kq = (lane & 7) * 4, mrow = mb * 32 + row;
    arow[i] = mrow < g.M ? (mrow * g.K + kq) * 4 : (int)0x80000000;
    awr[i] = kq * 33 + row;                    // LDS float index of element (kq, row); (kq + e, row) is e * 33 further
  }
  const int ard = hi * 33 + lo;                // LDS float index of element (hi, lo); MFMA j reads (2j + hi, lo) = ard + 66 j
  const int ntask = g.nblk * g.nsl;
  // tasks in the order (slice, block): every slice but the last is kc long, so the waves' first tasks are the long ones and the
  // short last slices fill up behind (or, with a wave per task, beside) them
  for (int task = wave; task < ntask; task += nwave) {
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
    ct_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    // stage: a step's four filter pieces -> this wave's LDS buffer (k-major), then its sixteen fragment elements back into registers.
    // Only this wave touches the buffer and a wave's LDS operations execute in order: no barrier.
    auto stage = [&](const ct_f32x4 (&a)[4], float (&af)[kCH]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int el = 0; el < 4; el++) abuf[awr[i] + 33 * el] = a[i][el];
#pragma unroll
      for (int j = 0; j < kCH; j++) af[j] = abuf[ard + 66 * j];
    };
    // steps of 32 k; the ring issues kDepth - 1 steps beyond the slice's last one (unconditional loads into registers nobody reads)
    const int nsteps = (kend - k0 + 2 * kCH - 1) / (2 * kCH);
    ct_f32x4 a[kDepth][4];
    float x[kDepth][kCH];
    float af[2][kCH];
    static_assert(kDepth % 2 == 0, "the fragment sets (and the table-entry sets) alternate with the step's parity");
    int s = 0;
    // In both forms a step n = s + d multiplies with its side work cut into sixteen groups, one behind each MFMA (they issue while
    // that MFMA runs): load j (and, in the first four groups, filter piece j) of step n + kDepth - 1; the filter block of step n + 1
    // (loaded kDepth - 2 steps ago) through LDS into the fragment registers the next step multiplies from -- its pieces written in
    // groups 0-3, its fragments read back in groups 6-13.  The scheduling barrier behind every group keeps the compiler from sorting
    // the groups back into phases, and pins the ISSUE ORDER of the loads: the vector-memory counter retires in order, so a step can
    // be waited for with the later ones in flight only if its loads really were issued first.
    if constexpr (kFast) {
      // The gather offset of lane (lo, hi) for k = 18 q + m (m = 2 pi + hi: pair pi of the 18-k period, channel 2 q + m / 9, tap
      // r = m % 9) is  q * 2 HW  +  [org + (m / 9) HW + kh W + kw]: the bracket -- nine values per lane, with "the tap does not exist
      // for this pixel" folded in as an out-of-range offset -- lives in voff[], rotated so that voff[i] belongs to pair
      // (phi + i) % 9 of the step being issued (a step is 16 pairs: the rotation advances by 7, register renaming inside the
      // unrolled round); q * 2 HW * 4 bytes is the instruction's scalar offset: three candidates per step, chosen per gather by
      // scalar compares.  Steps beyond the slice are clamped into the image / the filter (their values are never multiplied).
      const int cs2 = 2 * HW * 4, qmax = g.K / 18 - 1;
      const int k0s = __builtin_amdgcn_readfirstlane(k0), cs2s = __builtin_amdgcn_readfirstlane(cs2), qms = __builtin_amdgcn_readfirstlane(qmax);
      const int kas = __builtin_amdgcn_readfirstlane(g.K - 32);
      // (the filter pieces of the first steps go out before the offset registers are formed: their latency runs under that arithmetic)
#pragma unroll
      for (int d = 0; d < kDepth - 1; d++) {
        const int sa0 = min(k0s + 32 * d, kas) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) a[d][i] = __builtin_bit_cast(ct_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, arow[i], sa0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);
      int voff[9];
      {
        const int phi0 = (k0 >> 1) % 9;
#pragma unroll
        for (int i = 0; i < 9; i++) {
          int pi = phi0 + i;
          pi -= pi >= 9 ? 9 : 0;
          const int m = 2 * pi + hi, c2 = m >= 9 ? 1 : 0, r = m - 9 * c2, kh = (r * 11) >> 5, kw = r - 3 * kh;
          const int bad = __builtin_amdgcn_sbfe(inv, r, 1);                  // 0 (the tap exists) or -1
          voff[i] = ((org + c2 * HW + kh * g.W + kw) | bad) << 2;
        }
      }
      auto rot7 = [&]() __attribute__((always_inline)) {
        int nv[9];
#pragma unroll
        for (int i = 0; i < 9; i++) nv[i] = voff[(i + 7) % 9];
#pragma unroll
        for (int i = 0; i < 9; i++) voff[i] = nv[i];
      };
      // the scalars of the step m being issued: first pair P = k0 / 2 + 16 m = 9 Q + phi; gather j sits in channel pair
      // Q + [j >= tw] + [j >= tw + 9] (tw = 9 - phi); sa: the step's byte offset into the filter rows
#define CT_STEP_SCALARS(m)                                                                                         \
      const int P_ = (k0s >> 1) + 16 * (m), Q_ = P_ / 9, tw_ = 9 - (P_ - 9 * Q_);                                    \
      const int sq0_ = min(Q_, qms) * cs2s, dq1_ = min(Q_ + 1, qms) * cs2s - sq0_, dq2_ = min(Q_ + 2, qms) * cs2s - sq0_ - dq1_; \
      const int sa_ = min(k0s + 32 * (m), kas) * 4;
#define CT_GATHER(j) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, voff[(j) % 9],                \
                         sq0_ + ((j) >= tw_ ? dq1_ : 0) + ((j) >= tw_ + 9 ? dq2_ : 0), 0))
#pragma unroll
      for (int d = 0; d < kDepth - 1; d++) {
        CT_STEP_SCALARS(d)
        (void)sa_;
#pragma unroll
        for (int j = 0; j < kCH; j++) x[d][j] = CT_GATHER(j);
        rot7();
        __builtin_amdgcn_sched_barrier(0);
      }
      stage(a[0], af[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (task == wave) CT_STAMP(2);
#pragma unroll 1
      for (; s + kDepth <= nsteps; s += kDepth) {          // whole rounds of the ring: straight-line code
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
          CT_STEP_SCALARS(s + d + kDepth - 1)
          ct_f32x4 (&aI)[4] = a[(d + kDepth - 1) % kDepth];
          float (&xI)[kCH] = x[(d + kDepth - 1) % kDepth];
          const ct_f32x4 (&aS)[4] = a[(d + 1) % kDepth];
          float (&afS)[kCH] = af[(d + 1) & 1];
#pragma unroll
          for (int j = 0; j < kCH; j++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[d & 1][j], x[d][j], acc, 0, 0, 0);
            if (j < 4) {
              aI[j] = __builtin_bit_cast(ct_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, arow[j], sa_, 0));
#pragma unroll
              for (int el = 0; el < 4; el++) abuf[awr[j] + 33 * el] = aS[j][el];
            }
            xI[j] = CT_GATHER(j);
            if (j >= 6 && j < 14) {
              afS[2 * (j - 6)] = abuf[ard + 66 * (2 * (j - 6))];
              afS[2 * (j - 6) + 1] = abuf[ard + 66 * (2 * (j - 6) + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          rot7();
        }
      }
#undef CT_STEP_SCALARS
#undef CT_GATHER
    } else {
      // a step's table entries are READ from LDS one step before its loads are issued (a wave alone on its SIMD has nothing else to
      // cover an LDS read's ~100 cycles), the addresses and the loads follow from registers
      int2 e[2][kCH];
      auto tabread = [&](int ks, int2 (&ee)[kCH]) __attribute__((always_inline)) {
        const int2 *tk = tab + ks + hi;
#pragma unroll
        for (int j = 0; j < kCH; j++) ee[j] = tk[2 * j];
      };
      tabread(k0, e[0]);
#pragma unroll
      for (int d = 0; d < kDepth - 1; d++) {
        tabread(k0 + (d + 1) * 2 * kCH, e[(d + 1) & 1]);
        const int ks = k0 + d * 2 * kCH;
        const bool kin = ks + (lane & 7) * 4 < kend;                     // (kend % 4 == 0: a 16-byte piece is inside the slice or beyond it)
#pragma unroll
        for (int i = 0; i < 4; i++)
          a[d][i] = __builtin_bit_cast(ct_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? arow[i] + ks * 4 : (int)0x80000000, 0, 0));
#pragma unroll
        for (int j = 0; j < kCH; j++) {
          const int bad = __builtin_amdgcn_sbfe(inv, e[d & 1][j].y, 1);         // 0 (the tap exists) or -1
          x[d][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, ((org + e[d & 1][j].x) | bad) << 2, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stage(a[0], af[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (task == wave) CT_STAMP(2);
#pragma unroll 1
      for (; s + kDepth <= nsteps; s += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
          const int ksI = k0 + (s + d + kDepth - 1) * 2 * kCH;                       // the step being issued
          const int2 *tk = tab + k0 + (s + d + kDepth) * 2 * kCH + hi;               // the step whose table entries are read
          const bool kin = ksI + (lane & 7) * 4 < kend;
          ct_f32x4 (&aI)[4] = a[(d + kDepth - 1) % kDepth];
          float (&xI)[kCH] = x[(d + kDepth - 1) % kDepth];
          const int2 (&eI)[kCH] = e[(d + kDepth - 1) & 1];
          int2 (&eR)[kCH] = e[d & 1];
          const ct_f32x4 (&aS)[4] = a[(d + 1) % kDepth];
          float (&afS)[kCH] = af[(d + 1) & 1];
#pragma unroll
          for (int j = 0; j < kCH; j++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[d & 1][j], x[d][j], acc, 0, 0, 0);
            if (j < 4) {
              aI[j] = __builtin_bit_cast(ct_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? arow[j] + ksI * 4 : (int)0x80000000, 0, 0));
#pragma unroll
              for (int el = 0; el < 4; el++) abuf[awr[j] + 33 * el] = aS[j][el];
            }
            {
              const int bad = __builtin_amdgcn_sbfe(inv, eI[j].y, 1);
              xI[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, ((org + eI[j].x) | bad) << 2, 0, 0));
            }
            eR[j] = tk[2 * j];
            if (j >= 6 && j < 14) {
              afS[2 * (j - 6)] = abuf[ard + 66 * (2 * (j - 6))];
              afS[2 * (j - 6) + 1] = abuf[ard + 66 * (2 * (j - 6) + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // the last nsteps % kDepth steps (none at C4): their loads are in flight since the last whole round and nothing is left to issue
#pragma unroll
    for (int d = 0; d < kDepth - 1; d++) {
      if (s + d < nsteps) {
        if (s + d + 1 < nsteps) stage(a[(d + 1) % kDepth], af[(d + 1) & 1]);
#pragma unroll
        for (int j = 0; j < kCH; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[d & 1][j], x[d][j], acc, 0, 0, 0);
      }
    }
    if (task == wave) CT_STAMP(3);
    float *dst = ct_lds + ((size_t)(p * g.nblk + blk) * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; i++) dst[i * 64] = acc[i];
  }
  CT_STAMP(4);
  __syncthreads();
  CT_STAMP(5);
  // ordered fold + store: a wave takes half of a pixel block's accumulator rows (items = block x half, over the waves)
  float *out = g.out + (int64_t)b * g.bsC;
  for (int item = wave; item < 2 * g.nblk; item += nwave) {
    const int blk = item >> 1, h8 = (item & 1) * 8;
    const int pix = g.n_cut + blk * 32 + lo;
    float run[8];
#pragma unroll
    for (int i = 0; i < 8; i++) run[i] = ct_lds[((size_t)blk * 16 + h8 + i) * 64 + lane];       // S_0: the first slice WRITES (beta == 0)
    for (int p = 1; p < g.nsl; p++) {
      const float *sp = ct_lds + ((size_t)(p * g.nblk + blk) * 16 + h8) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 8; i++) run[i] = run[i] + sp[i * 64];                                 // C += S_p, ascending p (gemm.nim:150-158)
    }
    if (pix < g.npix) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = mb * 32 + (i & 3) + 8 * ((h8 + i) >> 2) + 4 * hi;
        if (row < g.M) out[(int64_t)row * g.npix + pix] = run[i];
      }
    }
  }
  CT_STAMP(6);
}

}  // namespace

#ifdef CT_TIMING
unsigned long long *g_ct_dbg = nullptr;
#endif

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
  // every slice a whole number of 32-k steps: the loop without vector address arithmetic (no tap table)
  const bool fast = a.K % 32 == 0 && a.K >= 64;
  // the tap table of the other form covers every k a step of the ring can name: a slice of n steps reads the entries of steps
  // 0 .. floor(n / kDepth) * kDepth + kDepth - 1 (the ring runs whole rounds and kDepth steps ahead), 2 * kCH entries each
  int64_t ktab = 0;
  for (int64_t p = 0; p < nsl && !fast; p++) {
    const int64_t k0 = p * kc, len = std::min<int64_t>(a.K, k0 + kc) - k0, nst = (len + 2 * kCH - 1) / (2 * kCH);
    ktab = std::max(ktab, k0 + ((nst + kDepth - 1) / kDepth * kDepth + kDepth) * 2 * kCH);
  }
  // a wave per task (pixel block x slice), four at least (one per SIMD), eight at most
  const int nwave = (int)std::min<int64_t>(8, std::max<int64_t>(4, nblk * nsl));
  const size_t lds = (size_t)nsl * nblk * 16 * 64 * sizeof(float) + (size_t)ktab * sizeof(int2) + (size_t)nwave * kAStep * sizeof(float);
  if (lds > ((size_t)64 << 10) || (int64_t)a.batch * mblks > 0x7fffffffLL) return hipErrorNotSupported;      // (longer reductions: the round-3 tail forms)
  if ((double)a.M * a.K * 4.0 >= 2147483648.0 || (double)Cin * a.cH * a.cW * 4.0 >= 2147483648.0) return hipErrorNotSupported;   // 31-bit byte offsets
  ConvTailArgs g;
  g.filt = a.A; g.img = a.B; g.out = a.C;
  g.bsB = a.bsB; g.bsC = a.bsC;
  g.M = (int32_t)a.M; g.K = (int32_t)a.K; g.H = a.cH; g.W = a.cW; g.oW = a.coW; g.npix = (int32_t)npix;
  g.pH = a.cpH; g.pW = a.cpW; g.n_cut = (int32_t)a.col0; g.nblk = (int32_t)nblk; g.nsl = (int32_t)nsl; g.kc = kc; g.ktab = (int32_t)ktab;
#ifdef CT_TIMING
  g.dbg = g_ct_dbg;
#endif
  if (fast) hipLaunchKernelGGL(conv3x3_tail_kernel<true>, dim3((unsigned)(a.batch * mblks)), dim3(64 * nwave), lds, s, g);
  else hipLaunchKernelGGL(conv3x3_tail_kernel<false>, dim3((unsigned)(a.batch * mblks)), dim3(64 * nwave), lds, s, g);
  return hipGetLastError();
}

}  // namespace laser_hip
