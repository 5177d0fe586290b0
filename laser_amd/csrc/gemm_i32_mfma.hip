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
//   1. limb_planes_tiled_kernel (limb_planes.h): strided int32 operand -> four int8 planes P_p[x][k], k-contiguous for BOTH
//      operands (B is transposed on the way, like pack_B), zero-padded to tile multiples, so the GEMM
//      kernel has no edge handling at all on its loads;
//   2. gemm_i8limb_kernel: 128x128 workgroup tile, 8 waves of 32x64, 64 k per LDS stage, double
//      buffered.  The planes are k-contiguous, so the stage image is lane-linear and is filled by
//      LDS-DMA (`global_load_lds_dwordx4`, no staging registers, no ds_write): rows are 64 B, the
//      16-B chunk c of row r sits at slot c ^ ((r>>2)&3) -- applied on the SOURCE address of the DMA
//      and on the fragment read -- which makes the 32-row `ds_read_b128` conflict-free without
//      padding.  Per 32-k step a wave issues 20 MFMAs from 12 ds_read_b128, B fragments one n-block
//      ahead (register double buffer, order pinned with sched_barrier); the 8 DMA pieces of the next
//      stage ride between MFMAs; four accumulator groups (one per power of 256) = 128 registers;
//      epilogue recombines G0 + (G1<<8) + (G2<<16) + (G3<<24), applies alpha/beta (wrapping) and
//      stores with the caller's strides.
// For K > 8192 the accumulators are folded into a running 32-bit sum every 8192 k so that
// |G_s| <= 4*8192*2^14 = 2^29 can never reach the int32 limit (no reliance on how the hardware treats
// accumulator overflow).
#include <type_traits>

#include "common.h"
#include "limb_planes.h"

namespace laser_hip {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

constexpr int IBM = 128, IBN = 128;  // workgroup tile
constexpr int IBKB = 64;             // k (bytes of each limb plane) per LDS stage = two 32-k MFMA steps
constexpr int ITHREADS = 512;
constexpr int IFOLD_K = 8192;
constexpr int IPLANE = IBM * IBKB;   // bytes of one limb plane of one operand in a stage (IBM == IBN)
constexpr int ISTAGE = 8 * IPLANE;   // 4 A planes + 4 B planes = 64 KiB; two stages = 128 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// ---- 1. limb planes: limb_planes.h (32 x 128 tiles through LDS, both HBM sides coalesced) ----------------------

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

template <bool FOLD>
__global__ void __launch_bounds__(ITHREADS, 2) gemm_i8limb_kernel(const I8Args g) {
  extern __shared__ __attribute__((aligned(16))) int8_t ismem[];

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

  // LDS-DMA assignment: a stage is 64 pieces of 1 KiB (16 rows x 64 B); wave w moves the 8 pieces
  // (= 8 row groups) of ONE plane: waves 0-3 the A planes, waves 4-7 the B planes.
  // lane -> row (lane>>2) of the group, slot lane&3; it fetches source chunk slot ^ swizzle(row).
  const int dplane = wave & 3;
  const bool dma_b = wave >= 4;
  const int drow = lane >> 2, dsrc = (lane & 3) ^ ((lane >> 4) & 3);
  const int8_t *dma_src = (dma_b ? g.Bp + dplane * g.planeB + (n0 + drow) * g.Kpad
                                 : g.Ap + dplane * g.planeA + (m0 + drow) * g.Kpad) + dsrc * 16;
  const int dma_dst = (dma_b ? 4 * IPLANE : 0) + dplane * IPLANE;  // + stage + group*1024 (wave-uniform)
  auto dma_piece = [&](int stage, int64_t k0, int grp) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((glb_void_t *)(dma_src + (int64_t)grp * 16 * g.Kpad + k0),
                                     (lds_void_t *)(ismem + stage * ISTAGE + dma_dst + grp * 1024), 16, 0, 0);
  };

  i32x16 acc[4][2];  // [power of 256][n block]
  i32x16 res[FOLD ? 2 : 1];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[s][n][r] = 0;
  if constexpr (FOLD) {
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) res[n][r] = 0;
  }
  auto combine = [&](int n, int r) __attribute__((always_inline)) -> uint32_t {
    return (uint32_t)acc[0][n][r] + ((uint32_t)acc[1][n][r] << 8) + ((uint32_t)acc[2][n][r] << 16) +
           ((uint32_t)acc[3][n][r] << 24);
  };
  auto fold = [&]() __attribute__((always_inline)) {
    if constexpr (FOLD) {
      asm volatile("; int8-limb accumulator fold" ::: "memory");
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          res[n][r] = (int)((uint32_t)res[n][r] + combine(n, r));
          acc[0][n][r] = acc[1][n][r] = acc[2][n][r] = acc[3][n][r] = 0;
        }
    }
  };

  // fragment addressing: lane (row/col = lo, k half = hi) reads 16 consecutive k bytes -- the SAME k
  // pattern for A and B, which is all an integer dot product needs.  slot = chunk ^ ((row>>2)&3).
  const int fsw = (lo >> 2) & 3;
  const int a_off = (wm0 + lo) * IBKB, b_off = 4 * IPLANE + (wn0 + lo) * IBKB;
  i32x4 fa[2][4], fb[2][4];
  auto ld_a = [&](const int8_t *st, int ks, int slot) __attribute__((always_inline)) {
    const int c = ((ks * 2 + hi) ^ fsw) * 16;
#pragma unroll
    for (int p = 0; p < 4; p++) fa[slot][p] = *reinterpret_cast<const i32x4 *>(st + a_off + p * IPLANE + c);
  };
  auto ld_b = [&](const int8_t *st, int ks, int n, int slot) __attribute__((always_inline)) {
    const int c = ((ks * 2 + hi) ^ fsw) * 16;
#pragma unroll
    for (int q = 0; q < 4; q++) fb[slot][q] = *reinterpret_cast<const i32x4 *>(st + b_off + n * 32 * IBKB + q * IPLANE + c);
  };

  const int nkt = (int)(g.Kpad / IBKB);
  constexpr int FOLD_TILES = IFOLD_K / IBKB;

  // prologue: stage 0 <- tile 0
#pragma unroll
  for (int grp = 0; grp < 8; grp++) dma_piece(0, 0, grp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one K-tile = 4 micro-steps (ks, n) of 10 MFMAs; B fragments are read one micro-step ahead, A
  // fragments one k-step ahead; two DMA pieces of the next stage ride in each micro-step.
  auto k_tile = [&](auto MORE_, int kt) __attribute__((always_inline)) {
    constexpr bool more = decltype(MORE_)::value;
    const int8_t *st = ismem + (kt & 1) * ISTAGE;
    const int nst = (kt + 1) & 1;
    const int64_t k1 = (int64_t)(kt + 1) * IBKB;
    ld_a(st, 0, 0);
    ld_b(st, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int ks = m >> 1, n = m & 1;
      if (m + 1 < 4) {
        ld_b(st, (m + 1) >> 1, (m + 1) & 1, (m + 1) & 1);
        if (((m + 1) & 1) == 0) ld_a(st, (m + 1) >> 1, ((m + 1) >> 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);  // reads of micro-step m+1 stay ahead of the MFMAs of micro-step m
      const int as = ks & 1, bs = m & 1;
#define LH_PROD(P, Q) acc[P + Q][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[as][P], fb[bs][Q], acc[P + Q][n], 0, 0, 0);
      // the 10 limb products with p+q <= 3, ordered so neighbours hit different accumulator groups
      LH_PROD(3, 0) LH_PROD(0, 0) LH_PROD(0, 1)
      if (more) dma_piece(nst, k1, 2 * m);
      LH_PROD(0, 2) LH_PROD(0, 3) LH_PROD(1, 0) LH_PROD(1, 1)
      if (more) dma_piece(nst, k1, 2 * m + 1);
      LH_PROD(1, 2) LH_PROD(2, 0) LH_PROD(2, 1)
#undef LH_PROD
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next stage must have landed (own DMA: vmcnt) and every wave must be done reading this one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  int kt = 0;
  int next_fold = FOLD ? FOLD_TILES : 0x7fffffff;
  for (;;) {
    const int stop = min(next_fold, nkt - 1);
    for (; kt < stop; kt++) k_tile(std::true_type{}, kt);
    if (kt == next_fold && kt < nkt) {
      fold();
      next_fold += FOLD_TILES;
      continue;
    }
    break;
  }
  if (kt < nkt) k_tile(std::false_type{}, kt);

  // epilogue: C = beta*C0 + alpha*res, all mod 2^32; beta == 0 never reads C
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int64_t row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int64_t col = n0 + wn0 + 32 * n + lo;
      if (row < g.M && col < g.N) {
        int32_t *p = g.C + row * g.rsC + col * g.csC;
        uint32_t sum = combine(n, r);
        if constexpr (FOLD) sum += (uint32_t)res[n][r];
        uint32_t v = (uint32_t)g.alpha * sum;
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
    return launch_limb_planes<int32_t>(dst, src, X, a.K, sx, sk, Xpad, Kpad, s);
  };
  hipError_t e = planes(Ap, a.A, a.M, a.rsA, a.csA, Mpad);
  if (e != hipSuccess) return e;
  e = planes(Bp, a.B, a.N, a.csB, a.rsB, Npad);
  if (e != hipSuccess) return e;
  constexpr size_t lds = 2 * ISTAGE;
  static_assert(lds <= 160 * 1024, "LDS budget");
  const bool need_fold = Kpad > IFOLD_K;
  auto kern = need_fold ? gemm_i8limb_kernel<true> : gemm_i8limb_kernel<false>;
  static PerDeviceOnce attr[2];  // per kernel variant, per device
  e = attr[need_fold].run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (e != hipSuccess) return e;
  I8Args g;
  g.Ap = Ap; g.Bp = Bp;
  g.planeA = Mpad * Kpad; g.planeB = Npad * Kpad; g.Kpad = Kpad;
  g.M = a.M; g.N = a.N;
  g.alpha = a.alpha; g.beta = a.beta;
  g.C = a.C; g.rsC = a.rsC; g.csC = a.csC;
  g.tiles_m = (int)(Mpad / IBM); g.tiles_n = (int)(Npad / IBN);
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(ITHREADS), lds, s, g);
  return hipGetLastError();
}

}  // namespace laser_hip
