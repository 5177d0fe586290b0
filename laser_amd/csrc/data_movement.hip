// laser_amd/csrc/data_movement.hip -- the HBM-bound kernels around the GEMM: physical transposes
// (laser/primitives/swapaxes.nim:16-112), im2col (benchmarks/convolution/conv2d_im2col.nim:42-88)
// and the strided->dense panel copy used by the pre-pack API (gemm_prepacked.nim:87-218).
// No arithmetic: the only goals are fully coalesced 16-B-per-lane HBM traffic on both sides and
// enough workgroups (>> 256) to cover all CUs.
#include "common.h"

namespace laser_hip {

std::atomic<int> g_im2col_band{0};   // option "im2col_band": output pixels per workgroup band of the im2col kernel (0 = 256 vectors)

// ---- batched 2-D transpose: dst[n][j][i] = src[n][i][j] -------------------------------------------
// 64x64 tile through LDS, 16-byte accesses on BOTH HBM sides (the reference writes contiguously and
// reads strided, swapaxes.nim:34-39; with LDS in between both sides are contiguous here):
//   read : thread (ty, tx) loads V consecutive elements (16 B) of src row r0+ty+16i -> LDS tile[row][col]
//   write: thread gathers V consecutive SOURCE ROWS of one source column from LDS (V scalar reads,
//          row stride 65/66 words => conflict-free), packs them and stores 16 B of dst row c0+col
// VEC16 needs NR, NC multiples of V and 16-B aligned bases; otherwise the scalar form below runs.
template <typename T, bool VEC16, int TR = 64, int TC = 64, bool NT = false>
__global__ void __launch_bounds__(256) transpose_batched_kernel(T *__restrict__ dst, const T *__restrict__ src,
                                                                int64_t NR, int64_t NC, int64_t tiles_c,
                                                                int64_t tiles_r, int64_t ld_src, int64_t ld_dst) {
  constexpr int V = 16 / sizeof(T);        // elements per 16-byte access: 4 (b32), 2 (b64), 8 (b16), 16 (b8)
  __shared__ T tile[TR][TC + 1];           // TR source rows x TC source columns (+1: conflict-free column reads)
  const int64_t bid = blockIdx.x;
  const int64_t tc = bid % tiles_c, tr = (bid / tiles_c) % tiles_r, n = bid / (tiles_c * tiles_r);
  const T *s = src + n * NR * ld_src;      // (row pitches: NC / NR for the dense batched form; the leading dimensions of a single
  T *d = dst + n * NC * ld_dst;            // matrix when a strided operand is packed for the GEMM kernels)
  const int64_t r0 = tr * TR, c0 = tc * TC;
  const int t = threadIdx.x;
  if constexpr (VEC16) {
    using VT = __attribute__((ext_vector_type(V))) T;
    {
      constexpr int TPR = TC / V;          // threads per source row
      constexpr int RPI = 256 / TPR;       // source rows per iteration
      const int tx = t % TPR, ty = t / TPR;
#pragma unroll
      for (int i = 0; i < TR / RPI; i++) {
        const int row = ty + RPI * i;
        const int64_t r = r0 + row, c = c0 + V * tx;
        if (r < NR && c < NC) {
          const VT *p = reinterpret_cast<const VT *>(s + r * ld_src + c);
          const VT q = NT ? __builtin_nontemporal_load(p) : *p;
#pragma unroll
          for (int e = 0; e < V; e++) tile[row][V * tx + e] = q[e];
        }
      }
    }
    __syncthreads();
    {
      constexpr int TPW = TR / V;          // threads per destination row (= source column)
      constexpr int CPI = 256 / TPW;       // destination rows per iteration
      const int tx = t % TPW, ty = t / TPW;
#pragma unroll
      for (int i = 0; i < TC / CPI; i++) {
        const int col = ty + CPI * i;      // source column = destination row
        const int64_t c = c0 + col, r = r0 + V * tx;
        if (c < NC && r < NR) {
          VT q;
#pragma unroll
          for (int e = 0; e < V; e++) q[e] = tile[V * tx + e][col];
          VT *p = reinterpret_cast<VT *>(d + c * ld_dst + r);
          if (NT)
            __builtin_nontemporal_store(q, p);
          else
            *p = q;
        }
      }
    }
  } else {
    static_assert(VEC16 || (TR == 64 && TC == 64), "the scalar form is 64x64");
    const int tx = t % 64, ty = t / 64;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
      if (r < NR && c < NC) tile[ty + 4 * i][tx] = s[r * ld_src + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
      if (r < NR && c < NC) d[c * ld_dst + r] = tile[tx][ty + 4 * i];
    }
  }
}


template <typename T, int TR, int TC, bool NT>
static hipError_t launch_transpose_v(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC, hipStream_t s, int64_t ld_src,
                                     int64_t ld_dst) {
  const int64_t tiles_r = (NR + TR - 1) / TR, tiles_c = (NC + TC - 1) / TC;
  const int64_t blocks = N * tiles_r * tiles_c;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((transpose_batched_kernel<T, true, TR, TC, NT>), dim3((unsigned)blocks), dim3(256), 0, s, (T *)dst,
                     (const T *)src, NR, NC, tiles_c, tiles_r, ld_src, ld_dst);
  return hipGetLastError();
}

// ld_src / ld_dst: row pitches of the source / destination (0 = dense: NC / NR); N > 1 only with dense pitches
template <typename T>
static hipError_t launch_transpose_t(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC, hipStream_t s, int64_t ld_src = 0,
                                     int64_t ld_dst = 0) {
  constexpr int V = 16 / sizeof(T);
  if (ld_src <= 0) ld_src = NC;
  if (ld_dst <= 0) ld_dst = NR;
  const bool vec = (NR % V == 0) && (NC % V == 0) && (ld_src % V == 0) && (ld_dst % V == 0) &&
                   ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  if (vec) {
    // Tile shape by problem size (scripts/probes/transpose_probe.hip, timed from C++ so that no interpreter sits between
    // launches; profiles/r03/transpose_probe_v1.jsonl): a problem of a few dozen microseconds wants MANY small workgroups
    // -- 16 source rows x 256 columns (1-KiB read segments, 64-B write segments): 4000 x 2000 f32 5.92 TB/s and 4096^2
    // 6.37 vs 5.57 / 5.67 with the 64 x 128 tile (a plain copy kernel of the same bytes: 6.3 / 6.7) -- while from ~100 us on
    // the write segments matter more: 32 x 256 (16384 x 8192: 5.27 vs 5.00; 8192^2: 5.01 vs 5.04; copy kernel 5.7).
    if (N * NR * NC <= ((int64_t)1 << 24)) return launch_transpose_v<T, 16, 256, false>(dst, src, N, NR, NC, s, ld_src, ld_dst);
    return launch_transpose_v<T, 32, 256, false>(dst, src, N, NR, NC, s, ld_src, ld_dst);
  }
  const int64_t tiles_r = (NR + 63) / 64, tiles_c = (NC + 63) / 64;
  const int64_t blocks = N * tiles_r * tiles_c;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((transpose_batched_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, (T *)dst, (const T *)src,
                     NR, NC, tiles_c, tiles_r, ld_src, ld_dst);
  return hipGetLastError();
}

// dst[c * ld_dst + r] = src[r * ld_src + c]: one matrix with row pitches (a transposed GEMM operand packed row-major)
hipError_t launch_transpose_pitched(void *dst, int64_t ld_dst, const void *src, int64_t ld_src, int64_t NR, int64_t NC, int elem_size,
                                    hipStream_t s) {
  if (NR <= 0 || NC <= 0) return hipSuccess;
  if (ld_src < NC || ld_dst < NR) return hipErrorInvalidValue;
  if (elem_size == 4) return launch_transpose_t<uint32_t>(dst, src, 1, NR, NC, s, ld_src, ld_dst);
  if (elem_size == 8) return launch_transpose_t<uint64_t>(dst, src, 1, NR, NC, s, ld_src, ld_dst);
  if (elem_size == 2) return launch_transpose_t<uint16_t>(dst, src, 1, NR, NC, s, ld_src, ld_dst);
  if (elem_size == 1) return launch_transpose_t<uint8_t>(dst, src, 1, NR, NC, s, ld_src, ld_dst);
  return hipErrorInvalidValue;
}

hipError_t launch_transpose_batched(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC,
                                    int elem_size, hipStream_t s) {
  if (N <= 0 || NR <= 0 || NC <= 0) return hipSuccess;
  if (elem_size == 4) return launch_transpose_t<uint32_t>(dst, src, N, NR, NC, s);
  if (elem_size == 8) return launch_transpose_t<uint64_t>(dst, src, N, NR, NC, s);
  // (swapaxes.nim:16-19 is generic in T: 2- and 1-byte elements -- float16 / int16 / int8 tensors -- on the same tile kernel, 8 / 16
  // elements per 16-byte access)
  if (elem_size == 2) return launch_transpose_t<uint16_t>(dst, src, N, NR, NC, s);
  if (elem_size == 1) return launch_transpose_t<uint8_t>(dst, src, N, NR, NC, s);
  return hipErrorInvalidValue;
}

// ---- im2col: [batch][C][H][W] -> [batch][C*kH*kW][oH*oW] -------------------------------------------
// Same index arithmetic as conv2d_im2col.nim:62-87: row = -pH + krow + oh*sH, col = -pW + kcol + ow*sW, zero outside the image.
// Generic in the element type like the reference's im2col*[T] (conv2d_im2col.nim:42-50); no arithmetic, so one instantiation per
// element SIZE (b32: float32 / int32, b64: float64 / int64).
//
// Band kernel (round 5; the workspace is kH*kW times the input, so this is a WRITE stream): one workgroup = one (image, channel)
// x one band of output pixels [P0, P1) -- P0 a multiple of the 16-byte vector, the bands a near-EQUAL split of oH*oW (the round-4
// kernel cut 1024-pixel chunks: at 56x56 a quarter of its workgroups held 64 pixels) -- and ALL kH*kW workspace rows of that band.
// The input rows the band touches are staged ONCE in LDS, zero rows above / below the image and pW zero columns either side
// included, so the expansion has no bounds checks; every (krow, kcol) row of the workspace is then written from LDS with 16-byte
// stores.  The input is read once from HBM (the old kernel gathered every element kH*kW times through L1 / L2).
// ROWVEC (unit column stride, oW a multiple of the vector, kW <= V + 1: the 3x3 / 5x5 stride-1 cases): a thread's V pixels lie in ONE
// output row and their taps of one kernel row are V + kW - 1 CONSECUTIVE elements of an LDS row whose pitch is a multiple of V, so two
// 16-byte LDS reads serve all kW stores of that kernel row (lane-consecutive 16-byte reads: no bank conflicts); otherwise one 4-byte
// (8-byte) LDS read per element -- measured round 5 on C4: 2.47 TB/s (round-4 kernel) -> 4.63 (element reads) -> see profiles/r05.
template <typename T, bool ROWVEC>
__global__ void __launch_bounds__(256) im2col_band_kernel(T *__restrict__ ws, const T *__restrict__ in, int chunks, int chunk_pix,
                                                          int H, int W, int kH, int kW, int oH, int oW, int pH, int pW, int sH,
                                                          int sW, int Wp, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *lds = reinterpret_cast<T *>(lds_raw);
  constexpr int V = 16 / sizeof(T);
  using VT = __attribute__((ext_vector_type(V))) T;
  const int64_t nc = blockIdx.x / chunks;          // n*C + c
  const int chunk = (int)(blockIdx.x - nc * chunks);
  const int npix = oH * oW;
  const int P0 = chunk * chunk_pix, P1 = min(npix, P0 + chunk_pix);
  if (P0 >= P1) return;
  const int oh_first = P0 / oW, oh_last = (P1 - 1) / oW;
  const int row_lo = oh_first * sH - pH;           // input row held by LDS row 0
  const int nrows = (oh_last - oh_first) * sH + kH;
  const T *img = in + nc * (int64_t)H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < nrows; r += 4) {
    const int row = row_lo + r;
    const bool row_ok = row >= 0 && row < H;
    const T *src = img + (int64_t)row * W - pW;
    for (int c = lane; c < Wp; c += 64) lds[r * Wp + c] = (row_ok && c >= pW && c < W + pW) ? src[c] : (T)0;
  }
  __syncthreads();
  const int nquads = (P1 - P0 + V - 1) / V;
  T *out0 = ws + nc * (int64_t)kH * kW * npix;
  for (int q = threadIdx.x; q < nquads; q += 256) {
    const int p = P0 + q * V;
    int oh = p / oW, ow = p - oh * oW;
    if constexpr (ROWVEC) {
      // (p + V <= npix and 16-byte stores are legal: oW % V == 0, checked by the launcher)
      const T *row0 = lds + (oh - oh_first) * sH * Wp + ow;      // 16-byte aligned: Wp % V == 0, ow % V == 0
      for (int krow = 0; krow < kH; krow++) {
        const VT lo = *reinterpret_cast<const VT *>(row0 + krow * Wp), hi = *reinterpret_cast<const VT *>(row0 + krow * Wp + V);
        T w[2 * V];
#pragma unroll
        for (int e = 0; e < V; e++) { w[e] = lo[e]; w[V + e] = hi[e]; }
        T *o = out0 + (int64_t)(krow * kW) * npix + p;
#pragma unroll
        for (int kcol = 0; kcol <= V; kcol++) {
          if (kcol < kW) {
            VT v;
#pragma unroll
            for (int e = 0; e < V; e++) v[e] = w[e + kcol];
            *reinterpret_cast<VT *>(o + (int64_t)kcol * npix) = v;
          }
        }
      }
    } else {
      int base[V];
#pragma unroll
      for (int e = 0; e < V; e++) {
        base[e] = (p + e < npix) ? (oh - oh_first) * sH * Wp + ow * sW : 0;
        if (++ow == oW) { ow = 0; ++oh; }
      }
      const bool full = vec_ok && p + V <= P1;
      for (int krow = 0; krow < kH; krow++)
        for (int kcol = 0; kcol < kW; kcol++) {
          const int off = krow * Wp + kcol;
          T *o = out0 + (int64_t)(krow * kW + kcol) * npix + p;
          VT v;
#pragma unroll
          for (int e = 0; e < V; e++) v[e] = lds[base[e] + off];
          if (full) {
            *reinterpret_cast<VT *>(o) = v;
          } else {
#pragma unroll
            for (int e = 0; e < V; e++)
              if (p + e < P1) o[e] = v[e];
          }
        }
    }
  }
}

// Fallback (an image so wide that kH input rows do not fit in LDS): one workgroup = 1024 consecutive output pixels of ONE workspace
// row (image, channel, kernel row, kernel col), V pixels per thread, gathered through the caches.
template <typename T>
__global__ void __launch_bounds__(256) im2col_gather_kernel(T *__restrict__ ws, const T *__restrict__ in, int chunks, int H, int W,
                                                            int kH, int kW, int oH, int oW, int pH, int pW, int sH, int sW,
                                                            int vec_ok) {
  constexpr int V = 16 / sizeof(T);
  using VT = __attribute__((ext_vector_type(V))) T;
  const int64_t wrow = blockIdx.x / chunks;        // ((n*C + c)*kH + krow)*kW + kcol
  const int chunk = (int)(blockIdx.x % chunks);
  const int kcol = (int)(wrow % kW);
  const int krow = (int)((wrow / kW) % kH);
  const int64_t nc = wrow / ((int64_t)kW * kH);    // n*C + c
  const T *img = in + nc * (int64_t)H * W;
  T *out = ws + wrow * (int64_t)oH * oW;
  const int npix = oH * oW;
  const int p0 = (chunk * 256 + (int)threadIdx.x) * V;
  if (p0 >= npix) return;
  VT v;
  int oh = p0 / oW, ow = p0 - oh * oW;
#pragma unroll
  for (int e = 0; e < V; e++) {
    const int row = -pH + krow + oh * sH, col = -pW + kcol + ow * sW;
    v[e] = (p0 + e < npix && row >= 0 && row < H && col >= 0 && col < W) ? img[(int64_t)row * W + col] : (T)0;
    if (++ow == oW) { ow = 0; ++oh; }
  }
  if (vec_ok && p0 + V <= npix) {
    *reinterpret_cast<VT *>(out + p0) = v;
  } else {
#pragma unroll
    for (int e = 0; e < V; e++)
      if (p0 + e < npix) out[p0 + e] = v[e];
  }
}

template <typename T>
static hipError_t launch_im2col_t(T *ws, int64_t oH, int64_t oW, const T *in, int64_t batch, int64_t C, int64_t H, int64_t W,
                                  int64_t kH, int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, hipStream_t s) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t ncs = batch * C, npix = oH * oW;
  if (ncs <= 0 || npix <= 0 || kH <= 0 || kW <= 0) return hipSuccess;
  if (npix > 0x7fffffffLL / 8 || H * W > 0x7fffffffLL || pH < 0 || pW < 0 || pH > 0xffff || pW > 0xffff || sH > 0xffff || sW > 0xffff)
    return hipErrorInvalidValue;
  const int vec_ok = (npix % V == 0) && ((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  // bands: ~1024 b32 / 512 b64 pixels each, equal to within one vector; shrunk while the staged input rows exceed 64 KiB of LDS
  const bool rowvec = vec_ok && sW == 1 && oW % V == 0 && kW <= V + 1;
  int64_t Wp = std::max<int64_t>(W + 2 * pW, (oW - 1) * sW + kW);
  if (rowvec) Wp = std::max<int64_t>((Wp + V - 1) / V * V, oW + V);      // 16-byte row pitch; a row's second vector read stays inside it
  // band length (profiles/r05/im2col_bands_v1.jsonl): many (image, channel) planes -> longer bands (6 KiB of every workspace row per
  // workgroup: C4 4.58 -> 5.16 TB/s); few planes -> 4 KiB bands so that the launch still has thousands of workgroups (the stride-2 case
  // of that sweep loses 20 - 45 % on the long bands); option "im2col_band" overrides (tuning sweeps)
  const int64_t band = g_im2col_band > 0 ? std::max<int64_t>(V, g_im2col_band) : (ncs >= 2048 ? 392 * V : 256 * V);
  int64_t chunks = (npix + band - 1) / band, chunk_pix = 0;
  size_t lds = 0;
  for (;; chunks *= 2) {
    chunk_pix = ((npix + chunks - 1) / chunks + V - 1) / V * V;
    const int64_t nrows = ((chunk_pix + oW - 1) / oW + 1) * sH + kH;     // (an upper bound: a band may start mid-row)
    lds = (size_t)nrows * Wp * sizeof(T);
    if (lds <= (size_t)64 << 10 || chunk_pix <= V) break;
  }
  chunks = (npix + chunk_pix - 1) / chunk_pix;
  if (lds <= ((size_t)64 << 10) && ncs * chunks <= 0x7fffffffLL && Wp <= 0x7fffffLL) {
    if (rowvec)
      hipLaunchKernelGGL((im2col_band_kernel<T, true>), dim3((unsigned)(ncs * chunks)), dim3(256), lds, s, ws, in, (int)chunks, (int)chunk_pix,
                         (int)H, (int)W, (int)kH, (int)kW, (int)oH, (int)oW, (int)pH, (int)pW, (int)sH, (int)sW, (int)Wp, vec_ok);
    else
      hipLaunchKernelGGL((im2col_band_kernel<T, false>), dim3((unsigned)(ncs * chunks)), dim3(256), lds, s, ws, in, (int)chunks, (int)chunk_pix,
                         (int)H, (int)W, (int)kH, (int)kW, (int)oH, (int)oW, (int)pH, (int)pW, (int)sH, (int)sW, (int)Wp, vec_ok);
    return hipGetLastError();
  }
  const int64_t rows = ncs * kH * kW, gchunks = (npix + 256 * V - 1) / (256 * V);
  if (rows * gchunks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(im2col_gather_kernel<T>, dim3((unsigned)(rows * gchunks)), dim3(256), 0, s, ws, in, (int)gchunks, (int)H, (int)W,
                     (int)kH, (int)kW, (int)oH, (int)oW, (int)pH, (int)pW, (int)sH, (int)sW, vec_ok);
  return hipGetLastError();
}

hipError_t launch_im2col(void *ws, int64_t oH, int64_t oW, const void *in, int64_t batch, int64_t C, int64_t H, int64_t W, int64_t kH,
                         int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, int elem_size, hipStream_t s) {
  if (elem_size == 4) return launch_im2col_t<uint32_t>((uint32_t *)ws, oH, oW, (const uint32_t *)in, batch, C, H, W, kH, kW, pH, pW, sH, sW, s);
  if (elem_size == 8) return launch_im2col_t<uint64_t>((uint64_t *)ws, oH, oW, (const uint64_t *)in, batch, C, H, W, kH, kW, pH, pW, sH, sW, s);
  return hipErrorInvalidValue;
}
hipError_t launch_im2col_f32(float *ws, int64_t oH, int64_t oW, const float *in, int64_t batch, int64_t C,
                             int64_t H, int64_t W, int64_t kH, int64_t kW, int64_t pH, int64_t pW,
                             int64_t sH, int64_t sW, hipStream_t s) {
  return launch_im2col(ws, oH, oW, in, batch, C, H, W, kH, kW, pH, pW, sH, sW, 4, s);
}

// ---- strided -> dense zero-padded panel image (pre-pack) -------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) pack_pad_kernel(T *__restrict__ dst, int64_t Rpad, int64_t Cpad,
                                                       const T *__restrict__ src, int64_t R, int64_t Cc,
                                                       int64_t rs, int64_t cs, int relu) {
  const int64_t total = Rpad * Cpad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / Cpad, c = e % Cpad;
    T x = (r < R && c < Cc) ? src[r * rs + c * cs] : (T)0;
    if (relu) x = x > (T)0 ? x : (T)0;      // a fused prologue materialised (operands the `_pre` kernels do not take)
    dst[e] = x;
  }
}

template <typename T>
hipError_t launch_pack_pad(T *dst, int64_t Rpad, int64_t Cpad, const T *src, int64_t R, int64_t Cc,
                           int64_t rs, int64_t cs, hipStream_t s, int relu) {
  const int64_t total = Rpad * Cpad;
  if (total <= 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pack_pad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, dst, Rpad, Cpad, src, R,
                     Cc, rs, cs, relu);
  return hipGetLastError();
}
template hipError_t launch_pack_pad<float>(float *, int64_t, int64_t, const float *, int64_t, int64_t, int64_t, int64_t, hipStream_t, int);
template hipError_t launch_pack_pad<double>(double *, int64_t, int64_t, const double *, int64_t, int64_t, int64_t, int64_t, hipStream_t, int);
template hipError_t launch_pack_pad<int32_t>(int32_t *, int64_t, int64_t, const int32_t *, int64_t, int64_t, int64_t, int64_t, hipStream_t, int);
template hipError_t launch_pack_pad<int64_t>(int64_t *, int64_t, int64_t, const int64_t *, int64_t, int64_t, int64_t, int64_t, hipStream_t, int);

// ---- rank-N strided copy: `forEachStrided d in dst, s in src: d = s` -----------------------------
// (laser/tensor/initialization.nim:42-110: deepCopy / copyFrom of non-contiguous tensors.)  HBM-bound.
// Host side first simplifies the iteration space: extent-1 dimensions are dropped and neighbouring dimensions
// that are contiguous on BOTH sides are merged (a sliced row-major tensor collapses to rank 1-2).  Then
//   * a pure 2-D transpose (source = a contiguous matrix read through swapped strides) goes to the transpose
//     kernel -- the one pattern where "one element per thread" would read with a huge stride;
//   * otherwise one workgroup copies 1024 consecutive elements of the innermost dimension of one outer "row":
//     the outer index is decomposed once per workgroup (uniform), lanes walk the inner dimension.
struct StridedCopyArgs {
  int64_t shape[kMaxRank], dstride[kMaxRank], sstride[kMaxRank];  // dimension rank-1 is the inner one
  int64_t inner, chunks;  // inner extent, workgroups per outer row
  int32_t rank;
};
template <typename T>
__global__ void __launch_bounds__(256) copy_strided_kernel(T *__restrict__ dst, const T *__restrict__ src,
                                                           StridedCopyArgs a) {
  const int64_t row = blockIdx.x / a.chunks, chunk = blockIdx.x - row * a.chunks;
  int64_t rem = row, so = 0, dof = 0;
#pragma unroll
  for (int d = kMaxRank - 2; d >= 0; d--) {
    if (d < a.rank - 1) {
      const int64_t q = rem / a.shape[d], i = rem - q * a.shape[d];
      so += i * a.sstride[d];
      dof += i * a.dstride[d];
      rem = q;
    }
  }
  const int64_t is = a.sstride[a.rank - 1], id = a.dstride[a.rank - 1];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int64_t i = chunk * 1024 + threadIdx.x + 256 * j;
    if (i < a.inner) dst[dof + i * id] = src[so + i * is];
  }
}
// inner dimension shorter than a workgroup's 1024 elements: a workgroup takes 1024 / P consecutive outer rows
// (P = inner extent rounded up to a power of two), lanes still walk the inner dimension; the outer index is
// decomposed per (thread, slot).
template <typename T>
__global__ void __launch_bounds__(256) copy_strided_rows_kernel(T *__restrict__ dst, const T *__restrict__ src,
                                                                StridedCopyArgs a, int64_t rows, int log2p) {
  const int64_t is = a.sstride[a.rank - 1], id = a.dstride[a.rank - 1];
  const int rows_per_wg = 1024 >> log2p;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int slot = threadIdx.x + 256 * j;
    const int64_t row = (int64_t)blockIdx.x * rows_per_wg + (slot >> log2p), i = slot & ((1 << log2p) - 1);
    if (row < rows && i < a.inner) {
      int64_t rem = row, so = 0, dof = 0;
#pragma unroll
      for (int d = kMaxRank - 2; d >= 0; d--) {
        if (d < a.rank - 1) {
          const int64_t q = rem / a.shape[d], ii = rem - q * a.shape[d];
          so += ii * a.sstride[d];
          dof += ii * a.dstride[d];
          rem = q;
        }
      }
      dst[dof + i * id] = src[so + i * is];
    }
  }
}
template <typename T>
hipError_t launch_copy_strided(T *dst, const int64_t *dstrides, const T *src, const int64_t *sstrides,
                               const int64_t *shape, int rank, hipStream_t s) {
  // drop extent-1 dimensions, merge dimension pairs (outer o, inner i) with stride_o == stride_i * extent_i on both sides
  int64_t sh[kMaxRank], ds[kMaxRank], ss[kMaxRank];
  int r = 0;
  int64_t total = 1;
  for (int d = 0; d < rank; d++) {
    total *= shape[d];
    if (shape[d] == 1) continue;
    if (r > 0 && ds[r - 1] == dstrides[d] * shape[d] && ss[r - 1] == sstrides[d] * shape[d]) {
      sh[r - 1] *= shape[d];
      ds[r - 1] = dstrides[d];
      ss[r - 1] = sstrides[d];
    } else {
      sh[r] = shape[d]; ds[r] = dstrides[d]; ss[r] = sstrides[d];
      r++;
    }
  }
  if (total == 0) return hipSuccess;
  if (r == 0) { sh[0] = 1; ds[0] = 1; ss[0] = 1; r = 1; }
  // dst[i][j] = src viewed with strides (1, ld): a physical transpose of a contiguous [sh1][sh0] matrix
  if (r == 2 && ds[1] == 1 && ds[0] == sh[1] && ss[0] == 1 && ss[1] == sh[0])
    return launch_transpose_batched(dst, src, 1, sh[1], sh[0], (int)sizeof(T), s);
  StridedCopyArgs a;
  a.rank = r;
  int64_t rows = 1;
  for (int d = 0; d < kMaxRank; d++) {
    a.shape[d] = d < r ? sh[d] : 1;
    a.dstride[d] = d < r ? ds[d] : 0;
    a.sstride[d] = d < r ? ss[d] : 0;
    if (d < r - 1) rows *= sh[d];
  }
  a.inner = sh[r - 1];
  if (a.inner < 1024 && r > 1) {  // short rows: several per workgroup
    int log2p = 0;
    while ((1 << log2p) < a.inner) log2p++;
    const int64_t nb = (rows + (1024 >> log2p) - 1) / (1024 >> log2p);
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
    a.chunks = 1;
    hipLaunchKernelGGL(copy_strided_rows_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, dst, src, a, rows, log2p);
    return hipGetLastError();
  }
  a.chunks = (a.inner + 1023) / 1024;
  const int64_t blocks = rows * a.chunks;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(copy_strided_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, a);
  return hipGetLastError();
}
// ---- ordered combine of per-slice partial products (slice-parallel GEMM, capi.cpp: gemm_slice_parallel) ----
// C[i,j] = (..((beta*C0 or 0) + alpha*W[0][i,j]) + alpha*W[1][i,j] ..) + alpha*W[nsl-1][i,j], unfused, ascending slice:
// exactly the sequence of Laser's pc loop (gemm.nim:150-158) with W[p] = the kc-slice product S_p.  HBM-bound.
template <typename E>
__global__ void __launch_bounds__(256) combine_slices_kernel(E *__restrict__ C, int64_t rsC, int64_t csC, const E *__restrict__ W,
                                                             int64_t M, int64_t N, int nsl, E alpha, E beta) {
  const int64_t total = M * N;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t i = idx / N, j = idx - i * N;
    E *p = C + i * rsC + j * csC;
    E run = (E)0;
    if (beta != (E)0) {  // beta == 0 never reads C
      run = *p;
      if (beta != (E)1) {
#pragma clang fp contract(off)
        run = run * beta;
      }
    }
    for (int q = 0; q < nsl; q++) {
#pragma clang fp contract(off)
      const E t = alpha * W[(int64_t)q * total + idx];
      run = run + t;
    }
    *p = run;
  }
}
template <typename E>
hipError_t launch_combine_slices(E *C, int64_t rsC, int64_t csC, const E *W, int64_t M, int64_t N, int nsl, E alpha, E beta,
                                 hipStream_t s) {
  const int64_t blocks = std::min<int64_t>((M * N + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(combine_slices_kernel<E>, dim3((unsigned)blocks), dim3(256), 0, s, C, rsC, csC, W, M, N, nsl, alpha, beta);
  return hipGetLastError();
}
template hipError_t launch_combine_slices<float>(float *, int64_t, int64_t, const float *, int64_t, int64_t, int, float, float, hipStream_t);
template hipError_t launch_combine_slices<double>(double *, int64_t, int64_t, const double *, int64_t, int64_t, int, double, double, hipStream_t);

template hipError_t launch_copy_strided<uint32_t>(uint32_t *, const int64_t *, const uint32_t *, const int64_t *, const int64_t *, int, hipStream_t);
template hipError_t launch_copy_strided<uint64_t>(uint64_t *, const int64_t *, const uint64_t *, const int64_t *, const int64_t *, int, hipStream_t);

}  // namespace laser_hip
