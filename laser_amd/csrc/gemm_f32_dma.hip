// laser_amd/csrc/gemm_f32_dma.hip -- float32 GEMM for the headline class (row-major A and B, whole 256x128x32 tiles), operand
// tiles brought into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write.
//
// STATUS: an experiment that is built, parity-tested and measured, and NOT the default (laser_hip_set_f32_dma(1) selects it).
// Why it was tried: the compile-time ablation of the register-staged main loop (scripts/ablate_exact.py,
// profiles/r02/ablate_exact_v1.jsonl; 8192^3, 256x128x32, laser-order) prices its parts -- MFMAs + fragment reads alone run at
// 148.7 TFLOP/s (0.945 of peak), the LDS stores of the staging path cost 6 % by themselves (134.1 -> 148.3 with stores and
// their waits gone, 140.0 with only the loads gone), the exposed part of the HBM latency another 4 %, the barrier nothing --
// and LDS-DMA has neither stores nor staging registers.  What it measures (dma_kernel_probe_v1.jsonl,
// dma_kernel_ablation_v1.jsonl): 131.5 TFLOP/s laser-order at 8192^3 against 133.5 for the register-staged kernel (133.3 vs
// 131.9 at 4096^3; fast 129.8 vs 139.5 for the staged 256x256 tile).  The six DMA requests per tile do cost less than the
// loads + stores they replace (4.4 % against 10 %), but the layout DMA forces on the k-contiguous operand costs as much
// again on the read side: with the requests removed this kernel runs 137.3, not 148.7.
//
// What DMA can and cannot lay out.  One instruction moves 64 x 16 bytes from 64 arbitrary global addresses to 1 KiB of
// CONTIGUOUS LDS (lane-linear).  So an operand lands in the orientation it has in memory:
//   A (row-major, k-contiguous): LDS rows [x][32 k] of 128 bytes, k in natural order; the 16-byte chunk c of row x sits at
//     slot c ^ ((x>>1)&7) (the swizzle is applied on the DMA's SOURCE address), which puts the 16 lanes of every
//     ds_read_b128 lane group ({0-3,12-15,20-27}, ...; all reading chunk c of 16 different rows) on 16 distinct 16-byte
//     slots.  An MFMA step j takes k = 2j from lanes 0-31 and k = 2j+1 from lanes 32-63 (that mapping IS the ascending-k
//     chain), so of the four k in a chunk a lane uses two -- element hi and 2+hi, picked with two v_cndmask -- and a chunk
//     read feeds two k-steps.  (The register-staged kernel's k-quad image stores (k0 k2 k4 k6 | k1 k3 k5 k7) so that a read
//     feeds four; DMA cannot permute inside its 16 bytes.)
//   B (row-major, n-contiguous): LDS rows [k][128 n] of 512 bytes = Laser's B~[k][jj] panel as is; a fragment is one
//     ds_read_b32 per (k-step, block): 32 consecutive words per half-wave, conflict-free without a swizzle.
// Per 64 MFMAs a wave issues 16 ds_read_b128 + 32 ds_read_b32 + 32 v_cndmask + 6 DMA pieces, against 16 ds_read_b128 +
// 12 ds_write_b64 + 6 global_load_dwordx4 + ~40 address / pack VALU in the register-staged form.
//
// Pipeline: ring of three 48 KiB stages (A 32 KiB + B 16 KiB).  Tile T is computed from slot T%3 while the six DMA pieces
// of tile T+2 go into slot (T-1)%3, free since the barrier of tile T-1.  ONE rendezvous per tile, before the first
// fragments of tile T+1 are read: `s_waitcnt vmcnt(6) lgkmcnt(0)` (this wave's pieces of T+1 have landed -- they were
// requested a whole tile before the six of T+2 that may still fly -- and its reads of slot T%3 are complete), `s_barrier`.
// Fragments are register double-buffered in groups of four k-steps as in the staged kernel.
//
// Arithmetic: identical to gemm_mfma_kernel.h -- the same MFMA chain per element, restarted every kc = 512 in laser-order
// mode with the slice sums folded in ascending order, the same unfused epilogue -- so results are bit-identical to the
// staged kernels and to the CPU restatement of the reference (tests compare them).
#include <type_traits>

#include "gemm_mfma_kernel.h"

namespace laser_hip {
namespace f32dma {

constexpr int BM = 256, BN = 128, BK = 32, THREADS = 512, NST = 3;
constexpr int A_BYTES = BM * BK * 4, B_BYTES = BK * BN * 4, STAGE = A_BYTES + B_BYTES;  // 32 + 16 KiB
constexpr int MB = 32, TM = 2, TN = 2;                                                   // wave tile 64 x 64
constexpr int NJ = BK / 2;                                                               // 16 MFMA k-steps per tile
constexpr int KGRP = 4, NG = NJ / KGRP;                                                  // fragment groups of four k-steps

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
using M_ = Mma<float>;
using Acc = M_::Acc;

template <bool EXACT, bool A1>
__global__ void __launch_bounds__(THREADS, 2) gemm_f32_dma_kernel(const GemmArgs<float> g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  // -- which C tile: XCD-aware bijective remap, then a grouped raster (4 x 8 patch of 256x128 tiles per XCD) --
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, xcd = bid % 8, loc = bid / 8, q = nwg / 8, r = nwg % 8;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 4;
  const int width = GROUP_M * g.tiles_n;
  const int first_m = (wgid / width) * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int pid_m = first_m + (wgid % width) % gsz;
  const int pid_n = (wgid % width) / gsz;
  const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN + g.col0;
  const int64_t bz = blockIdx.y;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, lo = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

  const float *Ab = g.A + bz * g.bsA + m0 * g.rsA;
  const float *Bb = g.B + bz * g.bsB + n0;
  float *Cb = g.C + bz * g.bsC;

  // -- DMA assignment: wave w moves A pieces 4w .. 4w+3 (8 rows of 128 B each) and B pieces 2w, 2w+1 (2 k-rows of 512 B) --
  // A piece p, lane l: row r = 8p + (l>>3), LDS slot l&7 <- global chunk (l&7) ^ ((r>>1)&7); (r>>1)&7 = ((p&1)*4 + (l>>4)) & 7.
  // Addresses are buffer offsets from the tile origin: the lane part (one VGPR per piece parity for A, one for B) plus a
  // wave-uniform part in an SGPR (piece row + k position) -- no 64-bit pointer per piece, no vector address arithmetic in
  // the loop.  (The launcher keeps every offset under 2 GiB.)
  auto uniform_rsrc = [](const float *p) __attribute__((always_inline)) {
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(bu), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ra_rs = uniform_rsrc(Ab), rb_rs = uniform_rsrc(Bb);
  const int rsA4 = (int)g.rsA * 4, rsB4 = (int)g.rsB * 4;  // row strides in bytes
  int voffA[2];
#pragma unroll
  for (int par = 0; par < 2; par++) voffA[par] = (lane >> 3) * rsA4 + 16 * ((lane & 7) ^ ((par * 4 + (lane >> 4)) & 7));
  const int voffB = (lane >> 5) * rsB4 + 16 * (lane & 31);
  int ka = 0, kb = 0;  // byte offsets of the tile being requested along k: A advances by BK floats, B by BK rows
  auto dma_op = [&](int slot, int o) __attribute__((always_inline)) {  // piece o of 6: the four A pieces first
    if (o < 4) {
      const int p = wave * 4 + o;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_rs, (lds_void_t *)(dsm + slot * STAGE + p * 1024), 16, voffA[o & 1], 8 * p * rsA4 + ka, 0, 0);
    } else {
      const int p = wave * 2 + (o - 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_rs, (lds_void_t *)(dsm + slot * STAGE + A_BYTES + p * 1024), 16, voffB, 2 * p * rsB4 + kb, 0, 0);
    }
  };
  auto dma_advance = [&]() __attribute__((always_inline)) {
    ka += BK * 4;
    kb += BK * rsB4;
  };

  Acc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][n][r] = 0.0f;
  const float alpha = g.alpha, beta = g.beta;
  auto c_ptr = [&](int i, int n, int r) __attribute__((always_inline)) -> float * {
    const int64_t row = m0 + wm0 + MB * i + M_::acc_row(r, lane);
    const int64_t col = n0 + wn0 + MB * n + M_::acc_col(lane);
    return Cb + row * g.rsC + col * g.csC;
  };
  // beta*C0 exactly as the reference's epilogues: beta == 0 -> 0 without reading C, beta == 1 -> C, else C*beta
  auto scaled_c0 = [&](int i, int n, int r) __attribute__((always_inline)) -> float {
    if (beta == 0.0f) return 0.0f;
    const float c0 = *c_ptr(i, n, r);
    return beta == 1.0f ? c0 : M_::mul(c0, beta);
  };
  Acc run[EXACT ? TM : 1][EXACT ? TN : 1];
  if constexpr (EXACT) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int n = 0; n < TN; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) run[i][n][r] = scaled_c0(i, n, r);
  }

  // -- fragments: groups of four k-steps, two register slots --
  // A: chunk c (k = 4c .. 4c+3) of row x at slot c ^ ((x>>1)&7); block bases are multiples of 32, so the swizzle is of lo.
  // B: word (k, n) at k*128 + n.
  const int fsw = (lo >> 1) & 7;
  const int a_row = (wm0 + lo) * 128;                    // byte offset of this lane's row in block 0 (block i: + 32*128)
  const int b_col = A_BYTES + (wn0 + lo) * 4 + hi * 512;  // byte offset of this lane's column, k = hi
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 ra[2][2][TM];        // [slot][chunk of the group][block]
  float fb[2][KGRP][TN];  // [slot][k-step of the group][block]
  auto ldgroup = [&](const unsigned char *st, int grp, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
      const int c = 2 * grp + c2;
#pragma unroll
      for (int i = 0; i < TM; i++) ra[slot][c2][i] = *reinterpret_cast<const f4 *>(st + a_row + i * (32 * 128) + ((c ^ fsw) << 4));
    }
#pragma unroll
    for (int u = 0; u < KGRP; u++) {
      const int j = grp * KGRP + u;  // lane's k = 2j + hi
#pragma unroll
      for (int n = 0; n < TN; n++) fb[slot][u][n] = *reinterpret_cast<const float *>(st + b_col + j * 1024 + n * 128);
    }
  };
  // the k this lane feeds to step u of the group: element hi (u even) or 2+hi (u odd) of chunk u/2
  auto a_of = [&](int slot, int u, int i) __attribute__((always_inline)) -> float {
    const f4 q = ra[slot][u >> 1][i];
    return (u & 1) ? (hi ? q[3] : q[2]) : (hi ? q[1] : q[0]);
  };

  const int nkt = (int)(g.K / BK);
  const int kc_tiles = EXACT ? (g.kc / BK) : 0;

  // -- prologue: tiles 0 and 1 requested; tile 0 landed; first fragments --
#pragma unroll
  for (int o = 0; o < 6; o++) dma_op(0, o);
  dma_advance();
  if (nkt > 1) {
#pragma unroll
    for (int o = 0; o < 6; o++) dma_op(1, o);
    dma_advance();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  ldgroup(dsm, 0, 0);
  int st = 0;  // ring slot of the tile being computed
  // the A values of the NEXT k-step are picked (v_cndmask) while the current step's MFMAs issue: a select right in front
  // of the MFMA that reads it costs hazard wait states
  float av_cur[TM], av_nxt[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) av_cur[i] = a_of(0, 0, i);

  // One K-tile.  MORE / MORE2: tile T+1 / T+2 exists (compile-time, so the body is one basic block); FOLD_: first tile of
  // a Laser kc slice -- each accumulator block is folded into `run` right before its first MFMA, which takes a zero C.
  auto k_tile = [&](auto MORE_, auto MORE2_, auto FOLD_) __attribute__((always_inline)) {
    constexpr bool more = decltype(MORE_)::value, more2 = decltype(MORE2_)::value;
    constexpr bool fold_here = EXACT && decltype(FOLD_)::value;
    const unsigned char *cur = dsm + st * STAGE;
    const int st1 = (st == 2) ? 0 : st + 1, st2 = (st == 0) ? 2 : st - 1;  // slots of T+1 and of T+2 (= T-1)
    const unsigned char *nxt = dsm + st1 * STAGE;
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      if (gi == NG - 1 && more) {
        // rendezvous: my pieces of T+1 have landed (the six of T+2 may still fly), my reads of this slot are complete
        if (more2)
          asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (gi + 1 < NG)
        ldgroup(cur, gi + 1, (gi + 1) & 1);
      else if (more)
        ldgroup(nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // reads of group g+1 stay ahead of the MFMAs of group g
#pragma unroll
      for (int u = 0; u < KGRP; u++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int n = 0; n < TN; n++) {
            const float av = av_cur[i];
            if (fold_here && gi == 0 && u == 0) {
              if constexpr (A1)
                run[EXACT ? i : 0][EXACT ? n : 0] = M_::fold_acc1(run[EXACT ? i : 0][EXACT ? n : 0], acc[i][n]);
              else
                run[EXACT ? i : 0][EXACT ? n : 0] = M_::fold_acc(run[EXACT ? i : 0][EXACT ? n : 0], acc[i][n], alpha);
              acc[i][n] = M_::mma(av, fb[0][0][n], Acc{});
            } else {
              acc[i][n] = M_::mma(av, fb[gi & 1][u][n], acc[i][n]);
            }
            // the six DMA pieces of tile T+2 ride behind MFMAs of the first three groups (one every eighth MFMA)
            if (more2 && gi < NG - 1) {
              const int slot_i = (gi * KGRP + u) * (TM * TN) + i * TN + n;  // 0 .. 47
#ifndef LH_DMA_POS
#define LH_DMA_POS 0
#endif
#if LH_DMA_POS == 0
              if (slot_i % 8 == 3) dma_op(st2, slot_i / 8);
#elif LH_DMA_POS == 1   // late in each group: clear of the fragment-read burst at the group's start
              if (slot_i % 16 == 9) dma_op(st2, 2 * (slot_i / 16));
              if (slot_i % 16 == 14) dma_op(st2, 2 * (slot_i / 16) + 1);
#elif LH_DMA_POS == 2   // ablation: no requests at all (timing only, results wrong)
#endif
            }
            if (i == 0 && n == 0) {  // behind the step's first MFMA: the selects of the next step (next group / next tile at the ends)
#pragma unroll
              for (int i2 = 0; i2 < TM; i2++) {
                if (u + 1 < KGRP)
                  av_nxt[i2] = a_of(gi & 1, u + 1, i2);
                else if (gi + 1 < NG)
                  av_nxt[i2] = a_of((gi + 1) & 1, 0, i2);
                else if (more)
                  av_nxt[i2] = a_of(0, 0, i2);
                else
                  av_nxt[i2] = 0.0f;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
        for (int i = 0; i < TM; i++) av_cur[i] = av_nxt[i];
      }
    }
    if (more2) dma_advance();
    st = st1;
  };
  // a slice boundary among the last two tiles folds in the open (rare: K not a multiple of 2 kc near the end)
  auto fold_open = [&]() __attribute__((always_inline)) {
    if constexpr (EXACT) {
      asm volatile("; laser-order slice fold" ::: "memory");
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int n = 0; n < TN; n++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            run[i][n][r] = M_::add(run[i][n][r], M_::mul(alpha, acc[i][n][r]));
            acc[i][n][r] = 0.0f;
          }
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  int kt = 0;
  int next_fold = (EXACT && kc_tiles > 0) ? kc_tiles : 0x7fffffff;
  for (;;) {
    const int stop = min(next_fold, nkt - 2);
    for (; kt < stop; kt++) k_tile(T_{}, T_{}, F_{});
    if (kt == next_fold && kt < nkt - 2) {
      k_tile(T_{}, T_{}, T_{});
      kt++;
      next_fold += kc_tiles;
      continue;
    }
    break;
  }
  // the last two tiles (no more requests)
  if (kt == next_fold && kt < nkt) {
    fold_open();
    next_fold += kc_tiles;
  }
  if (kt + 1 < nkt) {
    k_tile(T_{}, F_{}, F_{});
    kt++;
    if (kt == next_fold && kt < nkt) {
      fold_open();
      next_fold += kc_tiles;
    }
  }
  if (kt < nkt) k_tile(F_{}, F_{}, F_{});

  // -- epilogue: last (or only) slice, then store with the caller's C strides --
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float base;
        if constexpr (EXACT)
          base = run[i][n][r];
        else
          base = scaled_c0(i, n, r);
        *c_ptr(i, n, r) = M_::add(base, M_::mul(alpha, acc[i][n][r]));
      }
}

}  // namespace f32dma

std::atomic<int> g_f32_dma{0};  // knob (laser_hip_set_f32_dma): 1 = take eligible problems (default 0: the register-staged kernels measure the same or better)

// Takes plain (unfused) float32 problems whose A and B are row-major and 16-byte aligned and whose extents are whole
// 256 x 128 x 32 tiles, when there are enough tiles for the 256x128 configuration to be the heuristic's choice anyway;
// hipErrorNotSupported otherwise (the register-staged kernels handle everything).
hipError_t launch_gemm_f32_dma(const GemmArgs<float> &args, bool laser_order, hipStream_t s) {
  using namespace f32dma;
  const GemmArgs<float> &a = args;
  if (!g_f32_dma || a.bias != nullptr || a.act != 0 || a.batch < 1) return hipErrorNotSupported;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorNotSupported;
  if (a.csA != 1 || a.csB != 1 || a.M % BM || a.N % BN || a.K % BK) return hipErrorNotSupported;
  if (a.rsA % 4 || a.rsB % 4 || a.bsA % 4 || a.bsB % 4 || a.rsA < a.K || a.rsB < a.N) return hipErrorNotSupported;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) return hipErrorNotSupported;
  if (a.Mext != a.M || a.Next != a.N) return hipErrorNotSupported;  // (pre-packed panel images keep their own path)
  // 32-bit buffer offsets from the tile origin: a 256-row panel of A and a K-row panel of B stay under 2 GiB
  if ((double)BM * (double)a.rsA * 4.0 + (double)a.K * 4.0 >= 2.0e9 || (double)a.K * (double)a.rsB * 4.0 + 512.0 >= 2.0e9) return hipErrorNotSupported;
  const int64_t tiles = (a.M / BM) * (a.N / BN) * (int64_t)a.batch;
  if (tiles < 512 || a.K < 1024) return hipErrorNotSupported;
  const bool exact = laser_order && a.K > 512;
  const bool a1 = exact && a.alpha == 1.0f;
  g_last_split = 0;
  g_last_f32_cfg = 4;  // diagnostics: the same 256x128x32 tile geometry as configuration 4 of the register-staged kernels
  GemmArgs<float> g = a;
  g.tiles_m = (int)(a.M / BM);
  g.tiles_n = (int)(a.N / BN);
  g.kc = exact ? 512 : 0;
  g.col0 = 0;
  constexpr size_t lds = (size_t)NST * STAGE;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto launch = [&](auto kern, PerDeviceOnce &once) -> hipError_t {
    if (hipError_t e = once.run([&] {
          return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        e != hipSuccess)
      return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)a.batch, 1), dim3(THREADS), lds, s, g);
    return hipGetLastError();
  };
  static PerDeviceOnce o0, o1, o2;
  if (!exact) return launch(gemm_f32_dma_kernel<false, false>, o0);
  if (a1) return launch(gemm_f32_dma_kernel<true, true>, o1);
  return launch(gemm_f32_dma_kernel<true, false>, o2);
}

}  // namespace laser_hip
