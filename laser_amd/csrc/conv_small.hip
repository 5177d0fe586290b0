// laser_amd/csrc/conv_small.hip -- convolutions with FEW output channels (M = C_out <= 32) and a short reduction
// (K = C_in*kH*kW <= 256): the reference's own convolution benchmark, (16,3,224,224) (*) (20,3,3,3)
// (benchmarks/convolution/conv2d_bench.nim:130-170), is this class.
//
// As a GEMM the problem is M = 20 rows: a 64-row MFMA tile is 31 % full and the K loop is one tile long, so the tiled
// kernels are launch- and latency-bound (124 us for 73 MB of traffic).  It is an HBM stream: 9.6 MB of input, 63 MB of
// output, 851 MFLOP.  Here one lane owns PPT output pixels and all M channels of them: the filter bank (K x M floats) sits
// in LDS and is read as broadcasts, the input is read with lanes along the output row (coalesced; the kH*kW-fold reuse
// comes out of L1/L2), every output channel is stored as one coalesced row segment.  No im2col matrix, no LDS traffic for
// the image.
//
// Arithmetic: per output element the ascending-k fused multiply-add chain from +0, k = (c*kH + kh)*kW + kw
// (conv2d_im2col.nim:62-87 order), zero-padding taps multiplied in as zeros -- exactly what the matrix cores compute on the
// implicit-GEMM path for K <= kc = 512 (one slice: laser-order and one-chain modes coincide), so the results are
// bit-identical to it and to the CPU restatement the tests use (a v_fma_f32 chain == the f32 MFMA's chain, as in gemm_skinny.hip).
#include "common.h"

namespace laser_hip {

std::atomic<int> g_conv_direct{1};   // option "conv_direct": 0 = always the implicit-GEMM kernels

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
  // K <= 128: the matrix-core form above; longer reductions: the VALU form
  if (a.K <= 128) return launch_mfma(g, a.batch, s);
  if (a.M <= 8) return launch_small<8, 4>(g, a.batch, s);
  if (a.M <= 16) return launch_small<16, 4>(g, a.batch, s);
  if (a.M <= 24) return launch_small<24, 2>(g, a.batch, s);
  return launch_small<32, 2>(g, a.batch, s);
}

}  // namespace laser_hip
