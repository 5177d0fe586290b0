// laser_amd/csrc/gemm_i64_mfma.hip -- int64 GEMM, bit-exact mod 2^64, on the gfx950 int8 matrix cores.
//
// The reference has a real SIMD kernel for int64 (`c + a*b` with two's-complement wrap-around: mullo_epi64 + add,
// gemm_ukernel_avx512.nim:58-74; the generic fallback does the same per element).  gfx950 has no 64-bit integer
// matrix instruction and its VALU needs a multi-instruction 64x64 multiply, but -- exactly like the int32 path
// (gemm_i32_mfma.hip) -- arithmetic mod 2^64 decomposes over signed 8-bit limbs:
//
//     a   == sum_{p=0..7} s_p(a) * 256^p   (mod 2^64),   s_p in [-128, 127]   (balanced base 256)
//     a*b == sum_{p+q<=7} s_p(a) s_q(b) * 256^(p+q)       (mod 2^64)           (p+q >= 8 vanishes)
//
// so  sum_k a_ik b_kj == sum_{s=0..7} 256^s * G_s[i,j],  G_s = sum_{p+q=s} sum_k s_p(a_ik) s_q(b_kj): 36 int8 x int8 ->
// int32 products (`v_mfma_i32_32x32x32_i8`) per 32x32x32 block instead of 32768 64-bit multiply-adds.  Integer sums
// are associative mod 2^64, so any order is bit-exact.  K is processed in chunks of <= 8192 (one launch each,
// beta = 1 from the second on), which bounds |G_s| <= 8 * 8192 * 2^14 = 2^30: no int32 accumulator can overflow, so
// nothing here depends on how the hardware treats accumulator overflow.
//
// Structure:
//   1. limb_planes_tiled_kernel (limb_planes.h): strided int64 operand -> eight int8 planes P_p[x][k], k-contiguous for both operands (B is
//      transposed on the way, like pack_B, gemm_packing.nim:63-94), zero-padded to tile multiples;
//   2. gemm_i8limb64_kernel: 128x64 workgroup tile, 8 waves, ONE 32x32 block per wave -- the eight accumulator groups
//      (one per power of 256) are 128 registers, which is what caps the wave tile.  32 k per LDS stage (48 KiB: 8 A planes
//      + 8 B planes of 32-byte rows), a ring of three (a stage is requested two steps ahead), filled by LDS-DMA (`global_load_lds_dwordx4`); the 16-byte chunk c
//      of row r sits at slot c ^ ((r>>3)&1) -- applied on the DMA's source address and on the fragment read -- which puts
//      the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) on 16 distinct 16-byte slots (conflict-free,
//      no padding).  A k-step is three phases that keep all 64 fragment registers single-buffered:
//          X: A_hi x B_lo (10 MFMAs)   while the A_lo / B_hi fragments of this step are read
//          Y: A_lo x B_lo (16 MFMAs)   while the A_hi fragments of the NEXT step are read (dead since X)
//          Z: A_lo x B_hi (10 MFMAs)   while the B_lo fragments of the NEXT step are read (dead since Y)
//      with the barrier between X and Y (next stage landed; this stage fully read) and the six DMA pieces of step t+3
//      riding between the MFMAs of Y and Z into the ring slot step t just freed.
//      Epilogue: sum_s G_s << 8s in 64-bit (G_0..3 sign-extended; of G_4..7 only the low 32 - 8(s-4) bits survive the
//      shift), alpha / beta wrapping, strided store.
#include <type_traits>

#include "common.h"
#include "limb_planes.h"

namespace laser_hip {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

namespace i64mfma {
constexpr int BM = 128, BN = 64;     // workgroup tile: 4 x 2 waves of one 32x32 block
constexpr int BKB = 32;              // k (bytes of each limb plane) per LDS stage = one MFMA step
constexpr int THREADS = 512;
constexpr int KCHUNK = 8192;         // k per launch (accumulator range, see above)
constexpr int PLANE_A = BM * BKB;    // bytes of one A limb plane in a stage
constexpr int PLANE_B = BN * BKB;
constexpr int STAGE = 8 * (PLANE_A + PLANE_B);  // 48 KiB
constexpr int PIECES = STAGE / 1024;            // 1 KiB DMA pieces per stage: 32 of A, 16 of B
constexpr int PIECES_PER_WAVE = PIECES / 8;
}  // namespace i64mfma

typedef __attribute__((address_space(3))) void lds_void64_t;
typedef __attribute__((address_space(1))) const void glb_void64_t;

// ---- 1. limb planes: limb_planes.h (32 x 128 tiles through LDS, both HBM sides coalesced) ----------------------

// ---- 2. GEMM on the limb planes ------------------------------------------------------------------------
struct I8Args64 {
  const int8_t *Ap, *Bp;  // [8][Mpad][Kpad], [8][Npad][Kpad]
  int64_t planeA, planeB, Kpad;
  int64_t M, N;
  int64_t alpha, beta;
  int64_t *C;
  int64_t rsC, csC;
  int32_t tiles_m, tiles_n;
};

__global__ void __launch_bounds__(i64mfma::THREADS, 2) gemm_i8limb64_kernel(const I8Args64 g) {
  using namespace i64mfma;
  extern __shared__ __attribute__((aligned(16))) int8_t ismem64[];

  // XCD-aware bijective remap + grouped raster (same scheme as the f32 kernel)
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, xcd = bid % 8, loc = bid / 8, q = nwg / 8, r = nwg % 8;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 4;  // 4 x 8 patch of 128x64 tiles per XCD: 512 x 512
  const int width = GROUP_M * g.tiles_n;
  const int first_m = (wgid / width) * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int pid_m = first_m + (wgid % width) % gsz;
  const int pid_n = (wgid % width) / gsz;
  const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, lo = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;  // 4 x 2 waves, one 32x32 block each

  // LDS-DMA: a stage is 48 pieces of 1 KiB (32 rows x 32 B); wave w moves pieces 6w .. 6w+5.
  // lane -> row (lane>>1) of the piece, slot lane&1; it fetches source chunk slot ^ ((row>>3)&1).
  const int64_t lane_src = (int64_t)(lane >> 1) * g.Kpad + (((lane & 1) ^ ((lane >> 4) & 1)) * 16);
  auto dma_piece = [&](int stage, int64_t k0, int j) __attribute__((always_inline)) {
    const int idx = wave * PIECES_PER_WAVE + j;  // wave-uniform
    const int8_t *src;
    int dst;
    if (idx < 32) {
      const int plane = idx >> 2, rg = idx & 3;
      src = g.Ap + plane * g.planeA + (m0 + rg * 32) * g.Kpad;
      dst = plane * PLANE_A + rg * 1024;
    } else {
      const int plane = (idx - 32) >> 1, rg = (idx - 32) & 1;
      src = g.Bp + plane * g.planeB + (n0 + rg * 32) * g.Kpad;
      dst = 8 * PLANE_A + plane * PLANE_B + rg * 1024;
    }
    __builtin_amdgcn_global_load_lds((glb_void64_t *)(src + lane_src + k0), (lds_void64_t *)(ismem64 + stage * STAGE + dst), 16, 0, 0);
  };

  i32x16 acc[8];  // one per power of 256
#pragma unroll
  for (int s = 0; s < 8; s++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[s][r] = 0;

  // fragment addressing: lane (row/col = lo, k half = hi) reads 16 consecutive k bytes -- the same k pattern for A
  // and B, which is all an integer dot product needs.  slot = hi ^ ((row>>3)&1); block bases are multiples of 32.
  const int fslot = (hi ^ ((lo >> 3) & 1)) * 16;
  const int a_off = (wm0 + lo) * BKB + fslot, b_off = 8 * PLANE_A + (wn0 + lo) * BKB + fslot;
  i32x4 fa[8], fb[8];
  auto ld_a = [&](const int8_t *st, int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int p = p0; p < p0 + 4; p++) fa[p] = *reinterpret_cast<const i32x4 *>(st + a_off + p * PLANE_A);
  };
  auto ld_b = [&](const int8_t *st, int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = q0; q < q0 + 4; q++) fb[q] = *reinterpret_cast<const i32x4 *>(st + b_off + q * PLANE_B);
  };

  const int nkt = (int)(g.Kpad / BKB);

  // prologue: stages 0, 1, 2 in flight (ring of three: a stage is requested two whole steps before it is read, which
  // is what covers an HBM round trip -- with two buffers and one step of distance the kernel ran at 37 % of the matrix
  // rate, waiting on the DMA at every step), then the fragments phase X of step 0 needs
#pragma unroll
  for (int j = 0; j < PIECES_PER_WAVE; j++) dma_piece(0, 0, j);
  if (nkt > 1) {
#pragma unroll
    for (int j = 0; j < PIECES_PER_WAVE; j++) dma_piece(1, BKB, j);
  }
  if (nkt > 2) {
#pragma unroll
    for (int j = 0; j < PIECES_PER_WAVE; j++) dma_piece(2, 2 * BKB, j);
  }
  // stage 0 landed (the pieces of stages 1 and 2 may still fly)
  if (nkt > 2)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nkt > 1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ld_a(ismem64, 4);
  ld_b(ismem64, 0);

  int sb = 0;  // ring slot of the step being computed
#define LH_PROD(P, Q) acc[P + Q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[P], fb[Q], acc[P + Q], 0, 0, 0);
  // MORE: step kt+1 exists; MORE2: step kt+2 exists (its pieces are in flight at this step's barrier); MORE3: step kt+3
  // exists (its pieces are requested during this step, into the slot this step frees)
  auto k_step = [&](auto MORE_, auto MORE2_, auto MORE3_, int kt) __attribute__((always_inline)) {
    constexpr bool more = decltype(MORE_)::value, more2 = decltype(MORE2_)::value, more3 = decltype(MORE3_)::value;
    const int sb1 = (sb == 2) ? 0 : sb + 1;
    const int8_t *st = ismem64 + sb * STAGE;
    const int8_t *nx = ismem64 + sb1 * STAGE;
    const int64_t k3 = (int64_t)(kt + 3) * BKB;
    // -- X: A_hi x B_lo; the rest of this step's fragments are read meanwhile (neighbours hit different groups) --
    // (the reads go BEHIND the first two MFMAs: hipcc cannot count LDS operations across the loop's back edge, so the
    // first MFMA of a step waits for lgkmcnt(0) -- with this step's eight reads already issued that would be their
    // full latency; issued after it, the wait only covers the long-finished reads of the previous step)
    LH_PROD(7, 0) LH_PROD(6, 0)
    __builtin_amdgcn_sched_barrier(0);
    ld_a(st, 0);
    ld_b(st, 4);
    __builtin_amdgcn_sched_barrier(0);
    LH_PROD(6, 1) LH_PROD(5, 0) LH_PROD(5, 2)
    LH_PROD(5, 1) LH_PROD(4, 3) LH_PROD(4, 2) LH_PROD(4, 1) LH_PROD(4, 0)
    __builtin_amdgcn_sched_barrier(0);
    // this stage is fully read (by this wave) and the next one has landed (this wave's pieces; the six pieces of the
    // stage after next stay in flight): rendezvous
    if (more2)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (more)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) __syncthreads();
    // -- Y: A_lo x B_lo; A_hi of the next step is read, three DMA pieces of step kt+3 go into the slot just freed --
    if (more) ld_a(nx, 4);
    __builtin_amdgcn_sched_barrier(0);
    LH_PROD(0, 0) LH_PROD(0, 1) LH_PROD(0, 2) LH_PROD(0, 3)
    __builtin_amdgcn_sched_barrier(0);
    if (more3) dma_piece(sb, k3, 0);
    LH_PROD(1, 0) LH_PROD(1, 1) LH_PROD(1, 2) LH_PROD(1, 3)
    __builtin_amdgcn_sched_barrier(0);
    if (more3) dma_piece(sb, k3, 1);
    LH_PROD(2, 0) LH_PROD(2, 1) LH_PROD(2, 2) LH_PROD(2, 3)
    __builtin_amdgcn_sched_barrier(0);
    if (more3) dma_piece(sb, k3, 2);
    LH_PROD(3, 0) LH_PROD(3, 1) LH_PROD(3, 2) LH_PROD(3, 3)
    __builtin_amdgcn_sched_barrier(0);
    // -- Z: A_lo x B_hi; B_lo of the next step is read, the other three DMA pieces are issued --
    if (more) ld_b(nx, 0);
    __builtin_amdgcn_sched_barrier(0);
    LH_PROD(0, 7) LH_PROD(0, 6) LH_PROD(1, 6)
    __builtin_amdgcn_sched_barrier(0);
    if (more3) dma_piece(sb, k3, 3);
    LH_PROD(0, 5) LH_PROD(2, 5) LH_PROD(1, 5)
    __builtin_amdgcn_sched_barrier(0);
    if (more3) dma_piece(sb, k3, 4);
    LH_PROD(3, 4) LH_PROD(2, 4) LH_PROD(1, 4)
    __builtin_amdgcn_sched_barrier(0);
    if (more3) dma_piece(sb, k3, 5);
    LH_PROD(0, 4)
    __builtin_amdgcn_sched_barrier(0);
    sb = sb1;
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  int kt = 0;
  for (; kt < nkt - 3; kt++) k_step(T_{}, T_{}, T_{}, kt);
  if (kt < nkt - 2) {
    k_step(T_{}, T_{}, F_{}, kt);
    kt++;
  }
  if (kt < nkt - 1) {
    k_step(T_{}, F_{}, F_{}, kt);
    kt++;
  }
  if (kt < nkt) k_step(F_{}, F_{}, F_{}, kt);
#undef LH_PROD

  // epilogue: C = beta*C0 + alpha*sum_s (G_s << 8s), all mod 2^64; beta == 0 never reads C
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int64_t row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int64_t col = n0 + wn0 + lo;
    if (row < g.M && col < g.N) {
      int64_t *p = g.C + row * g.rsC + col * g.csC;
      uint64_t sum = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) sum += (uint64_t)(int64_t)acc[s][r] << (8 * s);  // exact (no overflow by construction)
#pragma unroll
      for (int s = 4; s < 8; s++) sum += (uint64_t)(uint32_t)acc[s][r] << (8 * s);  // bits above 63 vanish anyway
      uint64_t v = (uint64_t)g.alpha * sum;
      if (g.beta != 0) v += (uint64_t)g.beta * (uint64_t)*p;
      *p = (int64_t)v;
    }
  }
}

static inline int64_t rup64i(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

size_t gemm_i64_mfma_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  using namespace i64mfma;
  const int64_t Kpad = rup64i(K < KCHUNK ? K : KCHUNK, BKB);
  return (size_t)(8 * (rup64i(M, BM) + rup64i(N, BN)) * Kpad);
}

// `ws` must hold gemm_i64_mfma_workspace_bytes(M, N, K) bytes of device memory usable on stream s.
hipError_t launch_gemm_i64_mfma(const GemmArgs<int64_t> &a, void *ws, hipStream_t s) {
  using namespace i64mfma;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipSuccess;
  const int64_t Mpad = rup64i(a.M, BM), Npad = rup64i(a.N, BN);
  constexpr size_t lds = 3 * STAGE;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(PIECES_PER_WAVE * 8 == PIECES && PIECES_PER_WAVE == 6, "six DMA pieces per wave per stage");
  static PerDeviceOnce attr;
  hipError_t e = attr.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_i8limb64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (e != hipSuccess) return e;
  for (int64_t k0 = 0; k0 < a.K; k0 += KCHUNK) {
    const int64_t kc = (a.K - k0 < KCHUNK) ? a.K - k0 : KCHUNK;
    const int64_t Kpad = rup64i(kc, BKB);
    int8_t *Ap = (int8_t *)ws, *Bp = Ap + 8 * Mpad * Kpad;
    auto planes = [&](int8_t *dst, const int64_t *src, int64_t X, int64_t sx, int64_t sk, int64_t Xpad) {
      return launch_limb_planes<int64_t>(dst, src, X, kc, sx, sk, Xpad, Kpad, s);
    };
    e = planes(Ap, a.A + k0 * a.csA, a.M, a.rsA, a.csA, Mpad);
    if (e != hipSuccess) return e;
    e = planes(Bp, a.B + k0 * a.rsB, a.N, a.csB, a.rsB, Npad);
    if (e != hipSuccess) return e;
    I8Args64 g;
    g.Ap = Ap; g.Bp = Bp;
    g.planeA = Mpad * Kpad; g.planeB = Npad * Kpad; g.Kpad = Kpad;
    g.M = a.M; g.N = a.N;
    g.alpha = a.alpha; g.beta = (k0 == 0) ? a.beta : 1;  // later chunks add onto the first one's result
    g.C = a.C; g.rsC = a.rsC; g.csC = a.csC;
    g.tiles_m = (int)(Mpad / BM); g.tiles_n = (int)(Npad / BN);
    hipLaunchKernelGGL(gemm_i8limb64_kernel, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(THREADS), lds, s, g);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace laser_hip
