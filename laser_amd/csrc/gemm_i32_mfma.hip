// laser_amd/csrc/gemm_i32_mfma.hip -- int32 GEMM, bit-exact mod 2^32, on the gfx950 int8 matrix cores.
//
// Laser's integer kernels compute `c + a*b` with two's-complement wrap-around (mullo + add,
// gemm_ukernel_avx512.nim:40-41, gemm_ukernel_avx2.nim:10-11, gemm_ukernel_sse4_1.nim).  gfx950 has no
// 32-bit-integer MFMA, but arithmetic mod 2^32 decomposes exactly over signed 8-bit limbs:
//
//     a  ==  sum_{p=0..3} s_p(a) * 256^p   (mod 2^32),   s_p in [-128, 127]   (balanced base 256)
//     a*b == sum_{p+q<=3} s_p(a) s_q(b) * 256^(p+q)       (mod 2^32)           (p+q >= 4 vanishes)
//
// so  sum_k a_ik b_kj  ==  sum_{s=0..3} 256^s * G_s[i,j],   G_s = sum_{p+q=s} sum_k s_p(a_ik) s_q(b_kj),
// and every G_s is an int8 x int8 -> int32 matrix product: 10 `v_mfma_i32_32x32x32_i8` per 32x32x32
// block instead of 32768 VALU multiply-adds.  Integer sums are associative mod 2^32, so any order is
// bit-exact (the same argument the reference relies on across its ISA variants).
//
// Balanced digits in two VALU ops:  a'' = (a + 0x00808080) ^ 0x00808080 ; byte p of a'' is s_p as an
// int8 (adding 128 to the three low bytes with carry propagation, then flipping their sign bits; the
// top digit may be any representative mod 256 because 256^4 == 0).
//
// Structure (the GPU analogue of Laser's explicit packing pass, gemm_packing.nim:24-94):
//   1. limb_planes_kernel: strided int32 operand -> four int8 planes P_p[x][k], k-contiguous for BOTH
//      operands (B is transposed on the way, like pack_B), zero-padded to tile multiples, so the GEMM
//      kernel has no edge handling at all on its loads;
//   2. gemm_i8limb_kernel: 128x128 workgroup tile, 8 waves of 32x64, 64 k per LDS stage, double
//      buffered; per 32-k step a wave issues 20 MFMAs from 12 ds_read_b128; four accumulator groups
//      (one per power of 256) = 128 registers; epilogue recombines  G0 + (G1<<8) + (G2<<16) + (G3<<24),
//      applies alpha/beta (wrapping) and stores with the caller's strides.
// Accumulators are folded every 8192 k so that |G_s| <= 4*8192*2^14 = 2^29 can never reach the int32
// limit (no reliance on how the hardware treats accumulator overflow).
#include "common.h"

namespace laser_hip {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

constexpr int IBM = 128, IBN = 128;  // workgroup tile
constexpr int IBKB = 64;             // k (bytes of each limb plane) per LDS stage
constexpr int IROW = IBKB + 16;      // padded LDS row: 80 B => ds_read_b128 of 32 rows is conflict-free
constexpr int ITHREADS = 512;
constexpr int IFOLD_K = 8192;

// ---- 1. limb planes ---------------------------------------------------------------------------------
// planes[p][x][k] (int8), x < Xpad, k < Kpad; element (x, k) of the source at src[x*sx + k*sk].
// One thread = one x and 16 consecutive k -> one 16-byte store per plane.
__global__ void __launch_bounds__(256) limb_planes_kernel(int8_t *__restrict__ planes, const int32_t *__restrict__ src,
                                                          int64_t X, int64_t K, int64_t sx, int64_t sk, int64_t Xpad,
                                                          int64_t Kpad, int x_fast) {
  const int64_t kchunks = Kpad / 16;
  const int64_t total = Xpad * kchunks;
  const int64_t plane = Xpad * Kpad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    // lanes run along whichever source axis is contiguous so the 32-bit loads coalesce
    const int64_t x = x_fast ? e % Xpad : e / kchunks;
    const int64_t kq = x_fast ? e / Xpad : e % kchunks;
    uint32_t out[4][4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      uint32_t w[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int64_t k = kq * 16 + g * 4 + c;
        const uint32_t a = (x < X && k < K) ? (uint32_t)src[x * sx + k * sk] : 0u;
        w[c] = (a + 0x00808080u) ^ 0x00808080u;  // bytes = balanced base-256 digits of a
      }
      // 4x4 byte transpose: out[p][g] = { digit p of the 4 consecutive k }
      const uint32_t lo01 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), hi01 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
      const uint32_t lo23 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), hi23 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
      out[0][g] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
      out[1][g] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
      out[2][g] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
      out[3][g] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const i32x4 q = {(int)out[p][0], (int)out[p][1], (int)out[p][2], (int)out[p][3]};
      *reinterpret_cast<i32x4 *>(planes + p * plane + x * Kpad + kq * 16) = q;
    }
  }
}

// ---- 2. GEMM on the limb planes ------------------------------------------------------------------------
struct I8Args {
  const int8_t *Ap, *Bp;  // [4][Mpad][Kpad], [4][Npad][Kpad]
  int64_t planeA, planeB, Kpad;
  int64_t M, N;
  int32_t alpha, beta;
  int32_t *C;
  int64_t rsC, csC;
  int32_t tiles_m, tiles_n;
};

__global__ void __launch_bounds__(ITHREADS, 2) gemm_i8limb_kernel(const I8Args g) {
  extern __shared__ __attribute__((aligned(16))) int8_t ismem[];
  constexpr int STAGE = 4 * (IBM + IBN) * IROW;  // bytes per stage: 4 A planes then 4 B planes
  constexpr int BOFF = 4 * IBM * IROW;

  // XCD-aware bijective remap + grouped raster (same scheme as the f32 kernel)
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, xcd = bid % 8, loc = bid / 8, q = nwg / 8, r = nwg % 8;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * g.tiles_n;
  const int first_m = (wgid / width) * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int pid_m = first_m + (wgid % width) % gsz;
  const int pid_n = (wgid % width) / gsz;
  const int64_t m0 = (int64_t)pid_m * IBM, n0 = (int64_t)pid_n * IBN;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, lo = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;  // 4 x 2 waves, wave tile 32 x 64

  // staging: thread -> (row = t/4, 16-B chunk = t%4) of every limb plane of both operands
  const int srow = t >> 2, schunk = t & 3;
  const int8_t *ga = g.Ap + (m0 + srow) * g.Kpad + schunk * 16;
  const int8_t *gb = g.Bp + (n0 + srow) * g.Kpad + schunk * 16;
  const int soff = srow * IROW + schunk * 16;
  i32x4 ra[4], rb[4];
  auto gload = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ra[p] = *reinterpret_cast<const i32x4 *>(ga + p * g.planeA + k0);
      rb[p] = *reinterpret_cast<const i32x4 *>(gb + p * g.planeB + k0);
    }
  };
  auto sstore = [&](int st) __attribute__((always_inline)) {
    int8_t *base = ismem + st * STAGE;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      *reinterpret_cast<i32x4 *>(base + p * (IBM * IROW) + soff) = ra[p];
      *reinterpret_cast<i32x4 *>(base + BOFF + p * (IBN * IROW) + soff) = rb[p];
    }
  };

  i32x16 acc[4][2];  // [power of 256][n block]
  i32x16 res[2];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[s][n][r] = 0;
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) res[n][r] = 0;

  auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint32_t v = (uint32_t)acc[0][n][r] + ((uint32_t)acc[1][n][r] << 8) + ((uint32_t)acc[2][n][r] << 16) +
                           ((uint32_t)acc[3][n][r] << 24);
        res[n][r] = (int)((uint32_t)res[n][r] + v);
        acc[0][n][r] = acc[1][n][r] = acc[2][n][r] = acc[3][n][r] = 0;
      }
  };

  const int nkt = (int)(g.Kpad / IBKB);
  constexpr int FOLD_TILES = IFOLD_K / IBKB;
  int until_fold = FOLD_TILES;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int8_t *sA = ismem + (kt & 1) * STAGE;
    const int8_t *sB = sA + BOFF;
    const bool more = kt + 1 < nkt;
    if (more) gload((int64_t)(kt + 1) * IBKB);
#pragma unroll
    for (int ks = 0; ks < IBKB / 32; ks++) {
      // lane (row/col = lo, k half = hi) reads 16 consecutive k bytes: the SAME k pattern for A and B,
      // which is all an integer dot product needs
      const int koff = ks * 32 + hi * 16;
      i32x4 a[4];
#pragma unroll
      for (int p = 0; p < 4; p++) a[p] = *reinterpret_cast<const i32x4 *>(sA + p * (IBM * IROW) + (wm0 + lo) * IROW + koff);
      i32x4 b[2][4];
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          b[n][q] = *reinterpret_cast<const i32x4 *>(sB + q * (IBN * IROW) + (wn0 + 32 * n + lo) * IROW + koff);
        // the 10 limb products with p+q <= 3, ordered so consecutive MFMAs hit different accumulators
#define LH_PROD(P, Q)                                                                                             \
  acc[P + Q][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[P], b[0][Q], acc[P + Q][0], 0, 0, 0);                   \
  acc[P + Q][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[P], b[1][Q], acc[P + Q][1], 0, 0, 0);
      LH_PROD(0, 0) LH_PROD(0, 1) LH_PROD(0, 2) LH_PROD(0, 3) LH_PROD(1, 0)
      LH_PROD(1, 1) LH_PROD(1, 2) LH_PROD(2, 0) LH_PROD(2, 1) LH_PROD(3, 0)
#undef LH_PROD
    }
    if (--until_fold == 0) {
      until_fold = FOLD_TILES;
      fold();
    }
    if (more) sstore((kt + 1) & 1);
    __syncthreads();
  }
  fold();

  // epilogue: C = beta*C0 + alpha*res, all mod 2^32; beta == 0 never reads C
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int64_t row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int64_t col = n0 + wn0 + 32 * n + lo;
      if (row < g.M && col < g.N) {
        int32_t *p = g.C + row * g.rsC + col * g.csC;
        uint32_t v = (uint32_t)g.alpha * (uint32_t)res[n][r];
        if (g.beta != 0) v += (uint32_t)g.beta * (uint32_t)*p;
        *p = (int32_t)v;
      }
    }
}

static inline int64_t rup64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

size_t gemm_i32_mfma_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const int64_t Kpad = rup64(K, IBKB);
  return (size_t)(4 * (rup64(M, IBM) + rup64(N, IBN)) * Kpad);
}

// `ws` must hold gemm_i32_mfma_workspace_bytes(M, N, K) bytes of device memory usable on stream s.
hipError_t launch_gemm_i32_mfma(const GemmArgs<int32_t> &a, void *ws, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipSuccess;
  const int64_t Mpad = rup64(a.M, IBM), Npad = rup64(a.N, IBN), Kpad = rup64(a.K, IBKB);
  int8_t *Ap = (int8_t *)ws, *Bp = Ap + 4 * Mpad * Kpad;
  auto planes = [&](int8_t *dst, const int32_t *src, int64_t X, int64_t sx, int64_t sk, int64_t Xpad) {
    const int64_t total = Xpad * (Kpad / 16);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const int x_fast = (sx < 0 ? -sx : sx) < (sk < 0 ? -sk : sk);
    hipLaunchKernelGGL(limb_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, X, a.K, sx, sk, Xpad, Kpad, x_fast);
    return hipGetLastError();
  };
  hipError_t e = planes(Ap, a.A, a.M, a.rsA, a.csA, Mpad);
  if (e != hipSuccess) return e;
  e = planes(Bp, a.B, a.N, a.csB, a.rsB, Npad);
  if (e != hipSuccess) return e;
  constexpr size_t lds = 2 * 4 * (IBM + IBN) * IROW;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_done = false;
  if (!attr_done) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_i8limb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  I8Args g;
  g.Ap = Ap; g.Bp = Bp;
  g.planeA = Mpad * Kpad; g.planeB = Npad * Kpad; g.Kpad = Kpad;
  g.M = a.M; g.N = a.N;
  g.alpha = a.alpha; g.beta = a.beta;
  g.C = a.C; g.rsC = a.rsC; g.csC = a.csC;
  g.tiles_m = (int)(Mpad / IBM); g.tiles_n = (int)(Npad / IBN);
  hipLaunchKernelGGL(gemm_i8limb_kernel, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(ITHREADS), lds, s, g);
  return hipGetLastError();
}

}  // namespace laser_hip
