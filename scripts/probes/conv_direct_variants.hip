// scripts/probes/conv_direct_variants.hip -- where does the time of the small-channel direct convolution go?  The matrix-core
// kernel of laser_amd/csrc/conv_small.hip on the reference's bench geometry (16,3,224,224) (*) (20,3,3,3), with parts switched
// off: V0 full, V1 no stores, V2 no image loads, V3 neither (MFMA + LDS only), V4 plain 63 MB fill (dwordx4), V5 stores only
// in the kernel's pattern.  Probe only: nothing here ships.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
struct Args { const float *filt, *img; float *out; long bsB, bsC, rsC; int M, K, H, W, kH, kW, oW, npix; };
constexpr int CH = 16;
template <int V>
__global__ void __launch_bounds__(256) k(const Args g, int kpad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *s_af = lds; int *s_tap = (int *)(lds + 32 * kpad);
  const int t = threadIdx.x, lane = t & 63, lo = lane & 31, hi = lane >> 5, wave = t >> 6;
  const float *img = g.img + (long)blockIdx.y * g.bsB; float *out = g.out + (long)blockIdx.y * g.bsC;
  const int khw = g.kH * g.kW;
  for (int e = t; e < 32 * kpad; e += 256) { const int kk = e >> 5, m = e & 31; s_af[e] = (m < g.M && kk < g.K) ? g.filt[(long)m * g.K + kk] : 0.f; }
  for (int kk = t; kk < kpad; kk += 256) { const int kc = kk < g.K ? kk : 0; const int c = kc / khw, r = kc - c * khw, kh = r / g.kW, kw = r - kh * g.kW; s_tap[kk] = (c * g.H + kh) * g.W + kw; }
  __syncthreads();
  const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4, ng = (g.npix + 31) / 32;
  if (wid >= ng) return;
  const int nsteps = (ng - wid + nw - 1) / nw;
  const int dpix = 32 * nw, dq = dpix / g.oW, dr = dpix - dq * g.oW;
  int lpix = wid * 32 + lo, loh = lpix / g.oW, low = lpix - loh * g.oW, cpix = lpix;
  f32x16 acc;
  auto issue = [&](float (&x)[CH]) {
    const bool ok = lpix < g.npix;
    const float *src = img + (ok ? loh * g.W + low : 0);
#pragma unroll
    for (int j = 0; j < CH; j++) { if (V == 2 || V == 3 || V == 5) x[j] = (float)(lpix + j); else x[j] = src[s_tap[2 * j + hi]]; }
    lpix += dpix; loh += dq; low += dr; if (low >= g.oW) { low -= g.oW; loh++; }
  };
  auto compute = [&](const float (&x)[CH]) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    if (V == 5) { acc[0] = x[0]; }
    else {
#pragma unroll
    for (int j = 0; j < CH; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_af[(2 * j + hi) * 32 + lo], x[j], acc, 0, 0, 0);
    }
    const bool st = (V == 1 || V == 3) ? (acc[0] == 12345.678f) : true;
    if (cpix < g.npix && st) {
#pragma unroll
      for (int r = 0; r < 16; r++) { const int row = (r & 3) + 8 * (r >> 2) + 4 * hi; if (row < g.M) out[(long)row * g.rsC + cpix] = acc[r]; }
    }
    cpix += dpix;
  };
  float xa[CH], xb[CH];
  issue(xa);
  if (V == 7) {   // three register sets: the loads run two steps ahead of the arithmetic
    float xc[CH];
    issue(xb);
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 3) {
      issue(xc); compute(xa); if (s + 1 >= nsteps) break;
      issue(xa); compute(xb); if (s + 2 >= nsteps) break;
      issue(xb); compute(xc);
    }
    return;
  }
  if (V == 6) {
    float prev[16]; int ppix = g.npix;   // nothing to store yet
    auto compute6 = [&](const float (&x)[CH]) {
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < CH; j++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_af[(2 * j + hi) * 32 + lo], x[j], acc, 0, 0, 0);
        const int row = (j & 3) + 8 * (j >> 2) + 4 * hi;
        if (ppix < g.npix && row < g.M) out[(long)row * g.rsC + ppix] = prev[j];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) prev[r] = acc[r];
      ppix = cpix; cpix += dpix;
    };
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 2) { issue(xb); compute6(xa); if (s + 1 >= nsteps) break; issue(xa); compute6(xb); }
    if (ppix < g.npix) {
#pragma unroll
      for (int r = 0; r < 16; r++) { const int row = (r & 3) + 8 * (r >> 2) + 4 * hi; if (row < g.M) out[(long)row * g.rsC + ppix] = prev[r]; }
    }
    return;
  }
#pragma unroll 1
  for (int s = 0; s < nsteps; s += 2) { issue(xb); compute(xa); if (s + 1 >= nsteps) break; issue(xa); compute(xb); }
}
__global__ void fill(float4 *p, long n) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = make_float4(1, 2, 3, 4); }
template <int V> float run(const Args &a, int batch, int wgs) {
  const int kpad = 32; const size_t l = kpad * (32 * 4 + 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 300; i++) hipLaunchKernelGGL(k<V>, dim3(wgs, batch), dim3(256), l, 0, a, kpad);
  hipEventRecord(e0); for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k<V>, dim3(wgs, batch), dim3(256), l, 0, a, kpad);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 10.f;
}
int main() {
  const int batch = 16, C = 3, H = 224, W = 224, M = 20, oH = 222, oW = 222, npix = oH * oW, K = 27;
  float *img, *filt, *out; hipMalloc(&img, 4L * batch * C * H * W); hipMalloc(&filt, 4L * M * K); hipMalloc(&out, 4L * batch * M * npix + 4096);
  hipMemset(img, 0, 4L * batch * C * H * W); hipMemset(filt, 0, 4L * M * K);
  Args a{filt, img, out, (long)C * H * W, (long)M * npix, npix, M, K, H, W, 3, 3, oW, npix};
  for (int wgs : {8, 16, 24}) {
    printf("{\"wgs_per_image\": %d, \"V0_full_us\": %.1f, \"V1_nostore_us\": %.1f, \"V2_noload_us\": %.1f, \"V3_mfma_only_us\": %.1f, \"V5_store_only_us\": %.1f, \"V6_interleaved_stores_us\": %.1f, \"V7_three_sets_us\": %.1f}\n", wgs,
           run<0>(a, batch, wgs), run<1>(a, batch, wgs), run<2>(a, batch, wgs), run<3>(a, batch, wgs), run<5>(a, batch, wgs), run<6>(a, batch, wgs), run<7>(a, batch, wgs));
  }
  // padded planes: rsC a multiple of 32 floats -> every 128-byte store is line-aligned
  { Args b = a; b.rsC = (npix + 31) / 32 * 32; b.bsC = (long)M * b.rsC; float *o2; hipMalloc(&o2, 4L * batch * b.bsC + 4096); b.out = o2;
    printf("{\"aligned_planes\": 1, \"V0_full_us\": %.1f, \"V5_store_only_us\": %.1f}\n", run<0>(b, batch, 64), run<5>(b, batch, 64)); }
  const long n4 = (long)batch * M * npix / 4; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 100; i++) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float4 *)out, n4);
  hipEventRecord(e0); for (int i = 0; i < 100; i++) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float4 *)out, n4);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("{\"fill_63MB_us\": %.1f, \"fill_TBps\": %.2f}\n", ms * 10.f, 16.0 * n4 / (ms * 10.f) / 1e6);
  return 0;
}
