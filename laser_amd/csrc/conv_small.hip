// laser_amd/csrc/conv_small.hip -- convolutions with FEW output channels (M = C_out <= 32) and a short reduction
// (K = C_in*kH*kW <= 256): the reference's own convolution benchmark, (16,3,224,224) (*) (20,3,3,3)
// (benchmarks/convolution/conv2d_bench.nim:130-170), is this class.
//
// As a GEMM the problem is M = 20 rows: a 64-row MFMA tile is 31 % full and the K loop is one tile long, so the tiled
// kernels are launch- and latency-bound (124 us for 73 MB of traffic).  It is an HBM stream: 9.6 MB of input, 63 MB of
// output, 851 MFLOP.  Four forms, all with lanes along the output row (coalesced input reads whose kH*kW-fold reuse comes out
// of L1 / L2, one coalesced row segment per output channel), no im2col matrix, no LDS traffic for the image:
//   * 3x3 filters, M <= 24: the filter in SCALAR registers feeding packed FMAs (conv_direct_pairs_kernel when the output
//     width is even, the column stride 1 and nothing is padded -- the reference's bench shape, 22.9 us = 3.2 TB/s; else
//     conv_direct_scalar_kernel);
//   * other filters with K <= 128: one 32-row MFMA block per 32 pixels (conv_direct_mfma_kernel, 26.6 us on that shape);
//   * longer reductions: the filter as LDS broadcasts (conv_direct_small_kernel).
// Where the time goes on the bench shape (profiles/r04/conv_direct_dbg_v2.jsonl, batch 64 = three rounds of the chip): the
// kernel without its stores 41.9 us, its stores alone 33.6 us (= a fill of the output), both together 78.1 us -- the two
// do not overlap, with four or with eight waves per SIMD, with or without a software pipeline inside the wave; VALU busy 38 %.
//
// Arithmetic: per output element the ascending-k fused multiply-add chain from +0, k = (c*kH + kh)*kW + kw
// (conv2d_im2col.nim:62-87 order), zero-padding taps multiplied in as zeros -- exactly what the matrix cores compute on the
// implicit-GEMM path for K <= kc = 512 (one slice: laser-order and one-chain modes coincide), so the results are
// bit-identical to it and to the CPU restatement the tests use (a v_fma_f32 chain == the f32 MFMA's chain, as in gemm_skinny.hip).
#include "common.h"

namespace laser_hip {

std::atomic<int> g_conv_direct{1};   // option "conv_direct": 0 = always the implicit-GEMM kernels; 2 = without the scalar-filter forms

namespace {

struct ConvSmallArgs {
  const float *filt;   // [M][K]
  const float *img;    // [batch][Cin][H][W]
  float *out;          // [batch][M][npix] (row stride rsC)
  int64_t bsB, bsC, rsC;
  int32_t M, K, Cin, H, W, kH, kW, oW, npix, pH, pW, sH, sW;
};

// KHW: 0 = any filter size (runtime loops), 3 = 3x3 (the nine taps of a channel unrolled: their loads are issued together);
// PAD: false = no tap ever falls outside the image (no per-tap bounds test)
template <int MT, int PPT, int KHW, bool PAD>
__global__ void __launch_bounds__(256) conv_direct_small_kernel(const ConvSmallArgs g) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];   // [K][MT]: filter transposed, zero beyond M
  const int t = threadIdx.x;
  for (int idx = t; idx < g.K * MT; idx += 256) {
    const int k = idx / MT, m = idx % MT;
    wsm[idx] = m < g.M ? g.filt[(int64_t)m * g.K + k] : 0.0f;
  }
  __syncthreads();
  const float *img = g.img + (int64_t)blockIdx.y * g.bsB;
  int p[PPT], ih0[PPT], iw0[PPT], off[PPT];
  bool ok[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    p[j] = (int)blockIdx.x * (256 * PPT) + 256 * j + t;
    ok[j] = p[j] < g.npix;
    const int q = ok[j] ? p[j] : 0;              // (lanes past the image compute pixel 0 again and store nothing)
    const int oh = q / g.oW, ow = q - oh * g.oW;
    ih0[j] = oh * g.sH - g.pH;
    iw0[j] = ow * g.sW - g.pW;
    off[j] = ih0[j] * g.W + iw0[j];              // element offset of the window origin inside a channel plane (may be < 0)
  }
  float acc[PPT][MT];
#pragma unroll
  for (int j = 0; j < PPT; j++)
#pragma unroll
    for (int m = 0; m < MT; m++) acc[j][m] = 0.0f;
  const int HW = g.H * g.W;
  auto tap = [&](const float *plane, int kh, int kw, int k) __attribute__((always_inline)) {
    float x[PPT];
    const int toff = kh * g.W + kw;
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      if (PAD) {
        const bool in = (unsigned)(ih0[j] + kh) < (unsigned)g.H && (unsigned)(iw0[j] + kw) < (unsigned)g.W;
        x[j] = in ? plane[off[j] + toff] : 0.0f;
      } else {
        x[j] = plane[off[j] + toff];
      }
    }
    const float4 *wk = reinterpret_cast<const float4 *>(wsm + k * MT);
#pragma unroll
    for (int m4 = 0; m4 < MT / 4; m4++) {
      const float4 w = wk[m4];
#pragma unroll
      for (int j = 0; j < PPT; j++) {
        acc[j][4 * m4 + 0] = __builtin_fmaf(w.x, x[j], acc[j][4 * m4 + 0]);
        acc[j][4 * m4 + 1] = __builtin_fmaf(w.y, x[j], acc[j][4 * m4 + 1]);
        acc[j][4 * m4 + 2] = __builtin_fmaf(w.z, x[j], acc[j][4 * m4 + 2]);
        acc[j][4 * m4 + 3] = __builtin_fmaf(w.w, x[j], acc[j][4 * m4 + 3]);
      }
    }
  };
  // (An explicit software pipeline -- next channel's pixel values and the filter column two taps ahead in rotating register
  // sets -- was tried: the compiler spends 216-256 VGPRs on it and the kernel slows from 31 to 39-42 us on the reference's
  // bench shape; the plain loop below runs at 72 VGPRs with 6 waves per SIMD covering each other's latencies.)
  int k = 0;
  for (int c = 0; c < g.Cin; c++) {
    const float *plane = img + (int64_t)c * HW;
    if (KHW == 3) {
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) tap(plane, kh, kw, k + 3 * kh + kw);
      k += 9;
    } else {
      for (int kh = 0; kh < g.kH; kh++)
        for (int kw = 0; kw < g.kW; kw++, k++) tap(plane, kh, kw, k);
    }
  }
  float *out = g.out + (int64_t)blockIdx.y * g.bsC;
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    if (!ok[j]) continue;
#pragma unroll
    for (int m = 0; m < MT; m++)
      if (m < g.M) out[(int64_t)m * g.rsC + p[j]] = acc[j][m];
  }
}

// ---- 3x3 filters, at most 24 output channels: the VALU form with the filter in SCALAR registers -------------------------
// The LDS-filter kernel above reads every filter value as an LDS broadcast: at 20-24 channels x 2 pixels per lane the
// broadcasts, not the arithmetic, fill the LDS pipe; the matrix-core form below pays 32 rows for 20 channels and one VALU
// address computation per MFMA.  Here a filter value is a wave-uniform scalar load (s_load from the [M][K] bank as the caller
// holds it: no transposed copy, no LDS, no barrier) that feeds packed FMAs straight from its SGPR: a lane owns pairs of
// pixels in 64-bit registers (v_pk_fma_f32: two chains per instruction, the matrix cores' f32 rate) and all MT channels of
// them.  A row of the bank is addressed as min(m, M - 1): MT is M rounded up to a multiple of 4, the surplus chains are
// computed and dropped.  Same arithmetic as above: per output the ascending-k fused chain from +0.
// Measured on the reference's bench shape, (16,3,224,224) (*) (20,3,3,3): profiles/r04/conv_direct_probe_*.jsonl.
typedef __attribute__((ext_vector_type(2))) float cs_f32x2;
typedef __attribute__((address_space(4))) float cs_const_f32;

template <int MT, int PPT, bool PAD>
__global__ void __launch_bounds__(256, 4) conv_direct_scalar_kernel(const ConvSmallArgs g) {
  static_assert(PPT % 2 == 0, "pixels are held in pairs");
  const int t = threadIdx.x;
  // (the constant address space: wave-uniform loads from it are scalar loads whatever else the kernel does to memory)
  // grid z = groups of output channels, as in conv_direct_pairs_kernel
  const int zM = (g.M + (int)gridDim.z - 1) / (int)gridDim.z, m0 = (int)blockIdx.z * zM, Mz = min(zM, g.M - m0);
  const cs_const_f32 *filt = (const cs_const_f32 *)g.filt + (int64_t)m0 * g.K;
  const float *__restrict__ img = g.img + (int64_t)blockIdx.y * g.bsB;
  int p[PPT], ih0[PPT], iw0[PPT], off[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    p[j] = (int)blockIdx.x * (256 * PPT) + 256 * j + t;
    const int q = p[j] < g.npix ? p[j] : 0;      // (lanes past the image compute pixel 0 again and store nothing)
    const int oh = q / g.oW, ow = q - oh * g.oW;
    ih0[j] = oh * g.sH - g.pH;
    iw0[j] = ow * g.sW - g.pW;
    off[j] = ih0[j] * g.W + iw0[j];
  }
  cs_f32x2 acc[PPT / 2][MT];
#pragma unroll
  for (int j = 0; j < PPT / 2; j++)
#pragma unroll
    for (int m = 0; m < MT; m++) acc[j][m] = (cs_f32x2){0.0f, 0.0f};
  int wrow[MT];                                  // wave-uniform: element offset of channel m's filter row
#pragma unroll
  for (int m = 0; m < MT; m++) wrow[m] = (m < Mz ? m : Mz - 1) * g.K;
  const int HW = g.H * g.W;
  auto fetch = [&](const float *plane, int kh, int kw, float (&x)[PPT]) __attribute__((always_inline)) {
    const int toff = kh * g.W + kw;
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      if (PAD) {
        const bool in = (unsigned)(ih0[j] + kh) < (unsigned)g.H && (unsigned)(iw0[j] + kw) < (unsigned)g.W;
        x[j] = in ? plane[off[j] + toff] : 0.0f;
      } else {
        x[j] = plane[off[j] + toff];
      }
    }
  };
  int k = 0;
  for (int c = 0; c < g.Cin; c++) {
    const float *plane = img + (int64_t)c * HW;
    {
      // the nine taps of this channel in registers, then channel by channel: a filter row's nine values (one 8-dword and one
      // 1-dword scalar load) are pinned just before use and the next channel's are issued behind them, so few scalars are live
      // and a scalar load lands behind 9 * PPT / 2 packed FMAs (three taps at a time measured 25 % slower: the waits show).
      // Left alone the compiler hoists every scalar load of the loop body to its top and spills the scalars through vector
      // registers.
      float x[9][PPT];
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) fetch(plane, kh, kw, x[3 * kh + kw]);
      float wn[9];
#pragma unroll
      for (int q = 0; q < 9; q++) wn[q] = filt[wrow[0] + k + q];
#pragma unroll
      for (int m = 0; m < MT; m++) {
        float w[9];
#pragma unroll
        for (int q = 0; q < 9; q++) w[q] = wn[q];
        asm volatile("" : "+s"(w[0]), "+s"(w[1]), "+s"(w[2]), "+s"(w[3]), "+s"(w[4]), "+s"(w[5]), "+s"(w[6]), "+s"(w[7]), "+s"(w[8]));
        if (m + 1 < MT) {
#pragma unroll
          for (int q = 0; q < 9; q++) wn[q] = filt[wrow[m + 1] + k + q];
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
#pragma unroll
          for (int j = 0; j < PPT / 2; j++)
            acc[j][m] = __builtin_elementwise_fma((cs_f32x2){w[q], w[q]}, (cs_f32x2){x[q][2 * j], x[q][2 * j + 1]}, acc[j][m]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      k += 9;
    }
  }
  float *out = g.out + (int64_t)blockIdx.y * g.bsC + (int64_t)m0 * g.rsC;
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    if (p[j] >= g.npix) continue;
#pragma unroll
    for (int m = 0; m < MT; m++)
      if (m < Mz) out[(int64_t)m * g.rsC + p[j]] = acc[j / 2][m][j & 1];
  }
}

// ---- 3x3, unit column stride, no padding, an even output width (the reference's bench shape): pixel PAIRS --------------
// A lane owns PPL pairs of horizontally adjacent output pixels.  The two pixels' inputs for the taps kw = 0 and 2 of a filter
// row are one 16-byte load (elements o .. o+3: the pairs (0,1) and (2,3)), for kw = 1 one 8-byte load at o+1: two load
// instructions per filter row and pair instead of six, and every pair arrives in an even-aligned register pair as
// v_pk_fma_f32 wants it.  Addresses are a wave-uniform base (the channel plane + the filter row) plus one 32-bit byte offset
// per pair.  An output pair is one 8-byte store; lanes along the output: 512 contiguous bytes per wave and channel.
typedef __attribute__((ext_vector_type(4))) float cs_f32x4;
typedef __attribute__((ext_vector_type(4), aligned(4))) float cs_f32x4u;
typedef __attribute__((ext_vector_type(2), aligned(4))) float cs_f32x2u;

// CB = input channels whose loads are issued together in front of their arithmetic.  Shipped: 1.  CB = 3 (all of the bench
// shape's inputs requested at once, one exposed latency per wave instead of three; 86 registers) measured no faster: 25.1 vs
// 23.7 us per call at 16 images, 71.8 vs 72.3 at 64 (profiles/r04/conv_direct_probe_v6.jsonl).
// (Round 5, measured and not kept: two ADJACENT pairs per lane leaving as ONE 16-byte store per channel -- half the store
// instructions, 1 KiB per wave and channel, three / four waves per SIMD: 25.7 vs 22.2 us on the bench shape, 72.5 vs 66.5 at 64 images,
// 18.8 vs 17.9 with 16 channels, bit-identical -- profiles/r05/conv_small_quad_store_ab_v1.jsonl.  Wider stores are not what the
// kernel waits for; the eight-waves-per-SIMD form below stays.  Also measured and not kept: every workgroup walking 2 / 4 / 8 chunks
// (the stores of one chunk draining under the next chunk's arithmetic, the waves drifting out of their common compute-then-store
// phase): 24.3 / 31.0 / 43.7 us against 21.3 -- the time grows with the chunks per wave: a wave's own dependent path (loads, 540
// packed FMAs behind scalar filter loads, stores), not the chip's VALU or HBM rate, is what a launch of one round of waves takes;
// profiles/r05/conv_small_chunk_loop_ab_v1.jsonl.  And the same loop as a proper software pipeline -- loads(chunk i + 1) issued BEFORE
// the FMAs and stores of chunk i, so that the in-order vector-memory counter lets the stores drain under the next chunk's arithmetic;
// everything unconditional, stores through a bounds-checked buffer; ~110-135 registers, three waves per SIMD: 25.7 us (all channels
// per wave) / 23.2 (two channel groups) against 20.9 -- eight small waves per SIMD hide more than a hand-made pipeline in three;
// profiles/r05/conv_small_pipeline_ab_v1.jsonl.)
template <int MT, int PPL, int CB, int WPS>
__global__ void __launch_bounds__(256, WPS) conv_direct_pairs_kernel(const ConvSmallArgs g) {
  const int t = threadIdx.x;
  // grid z = groups of output channels (1 = all of them in one wave; 2 on launches of at most one round of waves: launch_scalar)
  const int zM = (g.M + (int)gridDim.z - 1) / (int)gridDim.z, m0 = (int)blockIdx.z * zM, Mz = min(zM, g.M - m0);
  const cs_const_f32 *filt = (const cs_const_f32 *)g.filt + (int64_t)m0 * g.K;
  const char *__restrict__ img = reinterpret_cast<const char *>(g.img + (int64_t)blockIdx.y * g.bsB);
  const int PW = g.oW >> 1, npairs = g.npix >> 1;
  int q[PPL];
  unsigned boff[PPL];
#pragma unroll
  for (int j = 0; j < PPL; j++) {
    q[j] = (int)blockIdx.x * (256 * PPL) + 256 * j + t;
    const int qq = q[j] < npairs ? q[j] : 0;     // (lanes past the image compute pair 0 again and store nothing)
    const int oh = qq / PW, pw = qq - oh * PW;
    boff[j] = (unsigned)((oh * g.sH) * g.W + 2 * pw) * 4u;
  }
  cs_f32x2 acc[PPL][MT];
#pragma unroll
  for (int j = 0; j < PPL; j++)
#pragma unroll
    for (int m = 0; m < MT; m++) acc[j][m] = (cs_f32x2){0.0f, 0.0f};
  int wrow[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) wrow[m] = (m < Mz ? m : Mz - 1) * g.K;
  const int64_t plane_bytes = (int64_t)g.H * g.W * 4, row_bytes = (int64_t)g.W * 4;
  for (int c0 = 0; c0 < g.Cin; c0 += CB) {
    // the raw 16-byte rows: elements 0,1 / 2,3 are the kw = 0 / 2 pairs as they lie, (1,2) is put together at its use (built
    // here it would be a wait for the load right behind its issue).  A channel past the last one re-reads the last (valid
    // addresses, a known number of loads) and is not used.
    cs_f32x4 raw[CB][3][PPL];
#pragma unroll
    for (int i = 0; i < CB; i++) {
      const int c = c0 + i < g.Cin ? c0 + i : g.Cin - 1;
      const char *plane = img + c * plane_bytes;
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int j = 0; j < PPL; j++) raw[i][kh][j] = *reinterpret_cast<const cs_f32x4u *>(plane + kh * row_bytes + boff[j]);
    }
#pragma unroll
    for (int i = 0; i < CB; i++) {
      if (c0 + i >= g.Cin) break;
      const int k = 9 * (c0 + i);
      cs_f32x2 x[9][PPL];
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int j = 0; j < PPL; j++) {
          x[3 * kh + 0][j] = (cs_f32x2){raw[i][kh][j][0], raw[i][kh][j][1]};
          x[3 * kh + 1][j] = (cs_f32x2){raw[i][kh][j][1], raw[i][kh][j][2]};
          x[3 * kh + 2][j] = (cs_f32x2){raw[i][kh][j][2], raw[i][kh][j][3]};
        }
      // channel by channel: a filter row's nine values (one 8-dword and one 1-dword scalar load) are pinned just before use and
      // the next channel's are issued behind them -- left alone the compiler hoists every scalar load of the body to its top
      // and spills the scalars through vector registers
      float wn[9];
#pragma unroll
      for (int r = 0; r < 9; r++) wn[r] = filt[wrow[0] + k + r];
#pragma unroll
      for (int m = 0; m < MT; m++) {
        float w[9];
#pragma unroll
        for (int r = 0; r < 9; r++) w[r] = wn[r];
        asm volatile("" : "+s"(w[0]), "+s"(w[1]), "+s"(w[2]), "+s"(w[3]), "+s"(w[4]), "+s"(w[5]), "+s"(w[6]), "+s"(w[7]), "+s"(w[8]));
        if (m + 1 < MT) {
#pragma unroll
          for (int r = 0; r < 9; r++) wn[r] = filt[wrow[m + 1] + k + r];
        }
#pragma unroll
        for (int r = 0; r < 9; r++) {
#pragma unroll
          for (int j = 0; j < PPL; j++) acc[j][m] = __builtin_elementwise_fma((cs_f32x2){w[r], w[r]}, x[r][j], acc[j][m]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float *out = g.out + (int64_t)blockIdx.y * g.bsC + (int64_t)m0 * g.rsC;
#pragma unroll
  for (int j = 0; j < PPL; j++) {
    if (q[j] >= npairs) continue;
#pragma unroll
    for (int m = 0; m < MT; m++)
      if (m < Mz) *reinterpret_cast<cs_f32x2u *>(out + (int64_t)m * g.rsC + 2 * q[j]) = acc[j][m];
  }
}

// ---- the same problem on the matrix cores: M <= 32 output channels are ONE 32-row MFMA block ------------------------
// v_mfma_f32_32x32x2_f32: lane (lo = l % 32, hi = l / 32) feeds A[lo][2j + hi] and B[2j + hi][lo].  The filter (A) never changes:
// its fragments sit in LDS in lane order (zero beyond M / K; a read is 64 consecutive words).  B[k][pixel] is the input value the
// tap k = (c, kh, kw) of output pixel `lo` reads: one global load per lane per MFMA, lanes along the output row (the kH*kW-fold
// reuse comes out of L1 / L2), no im2col matrix and no LDS staging of the image.  A wave walks GROUPS pixel groups of 32; the accumulator block holds
// 32 channels x 32 pixels, stored as one coalesced 128-byte row segment per channel and half wave.  k ascends through the MFMAs
// exactly as in the implicit-GEMM kernels (zero padding multiplied in as zeros), so the results are the same bits.
typedef __attribute__((ext_vector_type(16))) float cs_f32x16;

// LDS: filter fragments s_af[k][32 rows] (zero beyond M / K), then the tap table s_tap[k] = {input offset, kh << 16 | kw};
// k padded to a multiple of 2 * CH (one step = CH MFMAs).  The grid is sized to the machine, not to the problem: a wave strides
// through the 32-pixel groups of its image and keeps the loads of the NEXT step in flight while the matrix core works on this
// one (two register sets, the loop unrolled by two), so the filter / tap-table prologue is paid once per wave and the memory
// latency once per wave rather than once per group.
// CH = MFMAs per step: 16 in general; 8 / 12 / 14 when the whole reduction is that short (K = 27, the 3x3 RGB first layer, is
// 14 MFMAs instead of 16 -- the matrix core, not HBM, is the longer leg of this kernel).
template <bool PAD, int CH>
__global__ void __launch_bounds__(256) conv_direct_mfma_kernel(const ConvSmallArgs g, int kpad) {
  extern __shared__ __attribute__((aligned(16))) float cs_lds[];
  float *s_af = cs_lds;
  int2 *s_tap = reinterpret_cast<int2 *>(cs_lds + 32 * kpad);
  const int t = threadIdx.x, lane = t & 63, lo = lane & 31, hi = lane >> 5, wave = t >> 6;
  const float *img = g.img + (int64_t)blockIdx.y * g.bsB;
  float *out = g.out + (int64_t)blockIdx.y * g.bsC;
  const int khw = g.kH * g.kW;
  for (int e = t; e < 32 * kpad; e += 256) {
    const int k = e >> 5, m = e & 31;
    s_af[e] = (m < g.M && k < g.K) ? g.filt[(int64_t)m * g.K + k] : 0.0f;
  }
  for (int k = t; k < kpad; k += 256) {      // one integer division pair per tap and WORKGROUP
    const int kc = k < g.K ? k : 0;          // (k beyond K: A is zero there; any valid address will do for B)
    const int c = kc / khw, r = kc - c * khw, kh = r / g.kW, kw = r - kh * g.kW;
    s_tap[k] = make_int2((c * g.H + kh) * g.W + kw, (kh << 16) | kw);
  }
  __syncthreads();
  const int wid = (int)blockIdx.x * 4 + wave, nwaves = (int)gridDim.x * 4;
  const int ngroups = (g.npix + 31) / 32;
  if (wid >= ngroups) return;
  const int nch = kpad / (2 * CH);
  const int nsteps = (ngroups - wid + nwaves - 1) / nwaves * nch;
  // the load cursor (one step ahead of the arithmetic): the lane's pixel, its (oh, ow) carried along instead of divided out
  const int dpix = 32 * nwaves, dq = dpix / g.oW, dr = dpix - dq * g.oW;
  int lpix = wid * 32 + lo, loh = lpix / g.oW, low = lpix - loh * g.oW, lch = 0;
  int cpix = lpix, cch = 0;                  // the arithmetic cursor
  cs_f32x16 acc;
  auto issue = [&](float (&x)[CH]) {
    const bool ok = lpix < g.npix;
    const int ih0 = ok ? loh * g.sH - g.pH : 0, iw0 = ok ? low * g.sW - g.pW : 0;
    const float *src = img + (ih0 * g.W + iw0);
#pragma unroll
    for (int j = 0; j < CH; j++) {
      const int2 tap = s_tap[lch * 2 * CH + 2 * j + hi];
      if (PAD) {
        const bool in = (unsigned)(ih0 + (tap.y >> 16)) < (unsigned)g.H && (unsigned)(iw0 + (tap.y & 0xffff)) < (unsigned)g.W;
        x[j] = in ? src[tap.x] : 0.0f;
      } else {
        x[j] = src[tap.x];
      }
    }
    if (++lch == nch) {
      lch = 0; lpix += dpix; loh += dq; low += dr;
      if (low >= g.oW) { low -= g.oW; loh++; }
    }
  };
  auto compute = [&](const float (&x)[CH]) {
    if (cch == 0) {
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < CH; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_af[(cch * 2 * CH + 2 * j + hi) * 32 + lo], x[j], acc, 0, 0, 0);
    if (++cch == nch) {
      cch = 0;
      if (cpix < g.npix) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < g.M) out[(int64_t)row * g.rsC + cpix] = acc[r];
        }
      }
      cpix += dpix;
    }
  };
  float xa[CH], xb[CH];
  issue(xa);
#pragma unroll 1
  for (int s = 0; s < nsteps; s += 2) {
    // unconditional: past the last step the cursor points beyond npix and the lanes read (valid) addresses of pixel 0 -- a
    // branch round the loads would make the compiler wait for ALL outstanding loads before the first MFMA
    issue(xb);
    compute(xa);
    if (s + 1 >= nsteps) break;
    issue(xa);
    compute(xb);
  }
}

template <int CH>
hipError_t launch_mfma_ch(const ConvSmallArgs &g, int batch, hipStream_t s) {
  const int kpad = (g.K + 2 * CH - 1) / (2 * CH) * (2 * CH);
  const size_t lds = (size_t)kpad * (32 * sizeof(float) + sizeof(int2));
  // three workgroups per CU in one round (about 124 registers, four waves each), split evenly over the images (two per CU:
  // the same time without padding, 43 vs 37 us with the per-tap bounds tests of the padded form)
  const int64_t wgs_needed = (g.npix + 127) / 128, wgs_target = (256 * 3 + batch - 1) / batch;
  const dim3 grid((unsigned)std::max<int64_t>(1, std::min(wgs_needed, wgs_target)), (unsigned)batch);
  if (g.pH == 0 && g.pW == 0) hipLaunchKernelGGL((conv_direct_mfma_kernel<false, CH>), grid, dim3(256), lds, s, g, kpad);
  else hipLaunchKernelGGL((conv_direct_mfma_kernel<true, CH>), grid, dim3(256), lds, s, g, kpad);
  return hipGetLastError();
}

hipError_t launch_mfma(const ConvSmallArgs &g, int batch, hipStream_t s) {
  if (g.K <= 16) return launch_mfma_ch<8>(g, batch, s);
  if (g.K <= 24) return launch_mfma_ch<12>(g, batch, s);
  if (g.K <= 28) return launch_mfma_ch<14>(g, batch, s);
  return launch_mfma_ch<16>(g, batch, s);
}

template <int MT, int PPT>
hipError_t launch_small(const ConvSmallArgs &g, int batch, hipStream_t s) {
  const dim3 grid((unsigned)((g.npix + 256 * PPT - 1) / (256 * PPT)), (unsigned)batch);
  const size_t lds = (size_t)g.K * MT * sizeof(float);
  const bool k3 = g.kH == 3 && g.kW == 3;
  // no tap outside the image: no padding and the last window ends inside (true for every valid output shape)
  const bool nopad = g.pH == 0 && g.pW == 0;
  if (k3 && nopad) hipLaunchKernelGGL((conv_direct_small_kernel<MT, PPT, 3, false>), grid, dim3(256), lds, s, g);
  else if (k3) hipLaunchKernelGGL((conv_direct_small_kernel<MT, PPT, 3, true>), grid, dim3(256), lds, s, g);
  else hipLaunchKernelGGL((conv_direct_small_kernel<MT, PPT, 0, true>), grid, dim3(256), lds, s, g);
  return hipGetLastError();
}

template <int MT>
hipError_t launch_scalar_mt(const ConvSmallArgs &g, int batch, hipStream_t s, int zgroups = 1) {
  const dim3 grid((unsigned)((g.npix + 511) / 512), (unsigned)batch, (unsigned)zgroups);
  if (g.pH == 0 && g.pW == 0) hipLaunchKernelGGL((conv_direct_scalar_kernel<MT, 2, false>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((conv_direct_scalar_kernel<MT, 2, true>), grid, dim3(256), 0, s, g);
  return hipGetLastError();
}

template <int MT, int WPS>
hipError_t launch_pairs_mt(const ConvSmallArgs &g, int batch, hipStream_t s, int zgroups = 1) {
  const int npairs = g.npix / 2;
  const dim3 grid((unsigned)((npairs + 255) / 256), (unsigned)batch, (unsigned)zgroups);
  hipLaunchKernelGGL((conv_direct_pairs_kernel<MT, 1, 1, WPS>), grid, dim3(256), 0, s, g);
  return hipGetLastError();
}

// 3x3 filters, at most 24 output channels: the filter in scalar registers
hipError_t launch_scalar(const ConvSmallArgs &g, int batch, hipStream_t s) {
  // unit column stride, no padding, an even output width: pixel pairs (one pair per lane: 61 registers at 20 channels, eight
  // waves per SIMD -- 22.7 us on the reference's bench shape against 24.3 with two pairs per lane at four waves)
  if (g.pH == 0 && g.pW == 0 && g.sW == 1 && g.oW % 2 == 0 && (int64_t)g.H * g.W < ((int64_t)1 << 29)) {
    // Round 5: when the whole launch is at most ONE round of resident waves (8 per SIMD), the output channels go in two groups over
    // grid z -- a wave's own dependent path (27 packed FMAs and a store per channel behind scalar filter loads) is what such a launch
    // takes, so half the channels per wave and twice the waves: 22.1 -> 20.3 us on the reference's bench shape, 18.0 -> 16.6 with 16
    // channels; three groups, or two groups on launches of several rounds (64 images: 65.4 vs 64.8 us), gain nothing
    // (profiles/r05/conv_small_channel_groups_ab_v1.jsonl).  The input is read once more, from L2.
    const int64_t waves = (int64_t)((g.npix / 2 + 255) / 256) * batch * 4;
    if (g.M >= 12 && waves <= 256 * 4 * 8 && g_conv_direct != 3) {      // (option conv_direct = 3: never split -- the A/B switch)
      const int zM = (g.M + 1) / 2;
      switch ((zM + 3) / 4) {
        case 1: return launch_pairs_mt<4, 8>(g, batch, s, 2);
        case 2: return launch_pairs_mt<8, 8>(g, batch, s, 2);
        default: return launch_pairs_mt<12, 8>(g, batch, s, 2);
      }
    }
    switch ((g.M + 3) / 4) {
      case 1: return launch_pairs_mt<4, 8>(g, batch, s);
      case 2: return launch_pairs_mt<8, 8>(g, batch, s);
      case 3: return launch_pairs_mt<12, 8>(g, batch, s);
      case 4: return launch_pairs_mt<16, 8>(g, batch, s);
      case 5: return launch_pairs_mt<20, 8>(g, batch, s);
      default: return launch_pairs_mt<24, 6>(g, batch, s);
    }
  }
  {      // the same two-group split for launches of at most one round of waves (four waves per SIMD here)
    const int64_t waves = (int64_t)((g.npix + 511) / 512) * batch * 4;
    if (g.M >= 12 && waves <= 256 * 4 * 4 && g_conv_direct != 3) {
      const int zM = (g.M + 1) / 2;
      switch ((zM + 3) / 4) {
        case 1: return launch_scalar_mt<4>(g, batch, s, 2);
        case 2: return launch_scalar_mt<8>(g, batch, s, 2);
        default: return launch_scalar_mt<12>(g, batch, s, 2);
      }
    }
  }
  switch ((g.M + 3) / 4) {
    case 1: return launch_scalar_mt<4>(g, batch, s);
    case 2: return launch_scalar_mt<8>(g, batch, s);
    case 3: return launch_scalar_mt<12>(g, batch, s);
    case 4: return launch_scalar_mt<16>(g, batch, s);
    case 5: return launch_scalar_mt<20>(g, batch, s);
    default: return launch_scalar_mt<24>(g, batch, s);
  }
}

}  // namespace

// hipErrorNotSupported: not this kernel's class -- the caller takes the implicit-GEMM kernels
hipError_t launch_conv_direct_small_f32(const GemmArgs<float> &a, hipStream_t s) {
  if (!g_conv_direct) return hipErrorNotSupported;
  if (a.M < 1 || a.M > 32 || a.K < 1 || a.K > 256 || a.batch < 1 || a.batch > 65535) return hipErrorNotSupported;
  if (a.bias != nullptr || a.act != 0 || a.alpha != 1.0f || a.beta != 0.0f || a.col0 != 0 || a.cs_imgs != 0) return hipErrorNotSupported;
  if (a.csA != 1 || a.rsA != a.K || a.csC != 1 || a.ckH < 1 || a.ckW < 1 || a.csH < 1 || a.csW < 1) return hipErrorNotSupported;
  const int64_t khw = (int64_t)a.ckH * a.ckW;
  if (a.K % khw != 0 || a.coW < 1 || a.N % a.coW != 0) return hipErrorNotSupported;
  if ((double)a.cH * a.cW * (a.K / khw) >= 2.0e9 || a.N >= (int64_t)1 << 30) return hipErrorNotSupported;
  ConvSmallArgs g;
  g.filt = a.A; g.img = a.B; g.out = a.C;
  g.bsB = a.bsB; g.bsC = a.bsC; g.rsC = a.rsC;
  g.M = (int32_t)a.M; g.K = (int32_t)a.K; g.Cin = (int32_t)(a.K / khw);
  g.H = a.cH; g.W = a.cW; g.kH = a.ckH; g.kW = a.ckW; g.oW = a.coW; g.npix = (int32_t)a.N;
  g.pH = a.cpH; g.pW = a.cpW; g.sH = a.csH; g.sW = a.csW;
  // 3x3 filters with at most 24 output channels: the filter in scalar registers (option conv_direct = 2: never);
  // else K <= 128: the matrix-core form; longer reductions: the LDS-filter VALU form
  if (g_conv_direct != 2 && g.kH == 3 && g.kW == 3 && g.M <= 24) return launch_scalar(g, a.batch, s);
  if (a.K <= 128) return launch_mfma(g, a.batch, s);
  if (a.M <= 8) return launch_small<8, 4>(g, a.batch, s);
  if (a.M <= 16) return launch_small<16, 4>(g, a.batch, s);
  if (a.M <= 24) return launch_small<24, 2>(g, a.batch, s);
  return launch_small<32, 2>(g, a.batch, s);
}

}  // namespace laser_hip
