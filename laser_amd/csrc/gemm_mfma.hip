// laser_amd/csrc/gemm_mfma.hip -- host-side dispatch for the f32 / f64 MFMA GEMMs: picks the tile
// configuration and, per operand, the HBM->LDS "packing" loader that matches its strides.
// This replaces the reference's run-time ISA dispatch (gemm.nim:228-247) and its Tiles/partitionMNK
// geometry (gemm_tiling.nim:276-341) -- on the GPU the geometry is the workgroup tile.
#include <mutex>

#include "common.h"
#include "gemm_mfma_cfgs.h"

namespace laser_hip {

template <typename E>
struct CfgInfo {
  int bm, bn, bk, wm, wn, stages;
  bool vec, gen, exact;
  const char *name;
  hipError_t (*fn)(const GemmArgs<E> &, int, int, bool, hipStream_t);
};

#define X(IDX, BM, BN, BK, WM, WN, ST, OF, OE, WV, WG, WE) \
  {BM, BN, BK, WM, WN, ST, WV, WG, WE, #BM "x" #BN "x" #BK "_w" #WM "x" #WN "_s" #ST, &launch_gemm_f32_cfg<IDX>},
static const CfgInfo<float> kCfgsF32[LH_F32_NUM_CONFIGS] = {LH_F32_CONFIGS(X)};
#undef X
#define X(IDX, BM, BN, BK, WM, WN, ST, OF, OE, WV, WG, WE) \
  {BM, BN, BK, WM, WN, ST, WV, WG, WE, #BM "x" #BN "x" #BK "_w" #WM "x" #WN "_s" #ST, &launch_gemm_f64_cfg<IDX>},
static const CfgInfo<double> kCfgsF64[LH_F64_NUM_CONFIGS] = {LH_F64_CONFIGS(X)};
#undef X

int gemm_f32_config_count() { return LH_F32_NUM_CONFIGS; }
const char *gemm_f32_config_name(int cfg) {
  return (cfg >= 0 && cfg < LH_F32_NUM_CONFIGS) ? kCfgsF32[cfg].name : "?";
}

static inline int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

// Which loader can bring operand X (x = its M/N axis with stride sx, k with stride sk) into LDS?
// EPV = elements per 16 bytes (4 for f32, 2 for f64).
//   *vec  : plain 16-B vector loads  -- unit stride on one axis, other stride and batch stride
//           multiples of EPV elements, 16-B aligned base, no ragged tile in x or k;
//   *edge : the clamped / zero-selecting 16-B form -- same alignment, extents multiples of EPV only;
//   otherwise the predicated scalar loaders (GEN) handle anything.
template <typename E>
static int pick_mode(const E *p, int64_t sx, int64_t sk, int64_t bs, int64_t X, int64_t K, int bx, int bk,
                     bool *vec, bool *edge) {
  constexpr int64_t EPV = 16 / sizeof(E);
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (bs % EPV == 0);
  const bool full = (X % bx == 0) && (K % bk == 0);
  // *edge: 16-byte loads through a bounds-checked buffer descriptor -- unit stride along one direction, a
  // positive leading dimension, element alignment only, and the whole operand under 4 GB (32-bit offsets)
  const auto small = [&](int64_t ld) { return ld > 0 && ((X - 1) + (K - 1)) >= 0 &&
                                              (double)ld * (double)((sk == 1 ? X : K)) * sizeof(E) < 4.0e9; };
  *vec = *edge = false;
  if (sk == 1) {
    *vec = aligned && (sx % EPV == 0) && full;
    *edge = small(sx) && sx >= K;
    return LOAD_VEC_K;
  }
  if (sx == 1) {
    *vec = aligned && (sk % EPV == 0) && full;
    *edge = small(sk) && sk >= X;
    return LOAD_VEC_X;
  }
  return iabs64(sk) <= iabs64(sx) ? LOAD_VEC_K : LOAD_VEC_X;
}

static int to_gen(int mode) { return mode == LOAD_VEC_K ? LOAD_GEN_K : LOAD_GEN_X; }
static int to_edge(int mode) { return mode == LOAD_VEC_K ? LOAD_VEC_K_EDGE : LOAD_VEC_X_EDGE; }

template <typename E>
static int64_t tiles_of(const GemmArgs<E> &a, int bm, int bn) {
  return ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * (int64_t)a.batch;
}

// fp32: pick the configuration with the smallest predicted time.  Model (fitted on profiles/r01/
// sweep_f32_v8.json at 8192^3 and sweep_f32_small_v1.json at 512..3072):
//   a CU hosts up to R workgroups of a configuration (registers / LDS); with w of them resident it reaches
//   the fraction occ[w-1]/occ[R-1] of that configuration's full-residency speed (a lone 4-wave workgroup is
//   1 wave per SIMD and leaves most latency exposed);
//   tiles = q full rounds of 256*R + a tail that puts w = ceil(rem / 256) workgroups on the busiest CU:
//       time ~ (q*R + w / eff(w)) * bm*bn / speed(cfg)
// This is what makes 4100^3 take 128x128 tiles instead of 256x256 (2 rounds, the second 13 % full) and
// 1024^3..3072^3 take the 64x64 tiles (62 / 74 / 101 / 106 TFLOP/s measured vs 25 / 58 / 101 / 98 on 128x128).
constexpr int kCfgBig = 0, kCfgWide = 1, kCfgMid = 2, kCfgSmall = 3, kCfgWideExact = 4;
struct Cand {
  int cfg, bm, bn;
  double fast, laser, conv_fast, conv_laser;  // TFLOP/s at full residency; conv_*: with the gathering B loader (C4)
  int r_fast, r_laser;                        // co-resident workgroups per CU
  double occ[4];                              // relative CU throughput with 1..4 workgroups resident
  bool gen;
};
static const Cand kCands[] = {
    {kCfgBig, 256, 256, 141.5, 0.0, 124.7, 0.0, 1, 1, {1.0, 1.0, 1.0, 1.0}, false},
    {kCfgWideExact, 256, 128, 136.1, 133.8, 117.0, 0.0, 1, 1, {1.0, 1.0, 1.0, 1.0}, false},  // (laser-order gather would spill: cfg 1)
    {kCfgWide, 256, 128, 134.0, 131.1, 116.0, 117.0, 2, 1, {0.95, 1.0, 1.0, 1.0}, false},
    {kCfgMid, 128, 128, 136.4, 130.0, 119.0, 114.0, 3, 2, {0.755, 0.93, 1.0, 1.0}, true},
    {kCfgSmall, 64, 64, 120.8, 118.8, 99.0, 98.0, 4, 3, {0.45, 0.75, 0.92, 1.0}, true},
};
// predicted time of `tiles` tiles of candidate c, in units of (tile area / TFLOP/s): seconds = units * 2K * 256 / 1e12
static double cand_time(const Cand &c, int64_t tiles, bool exact, bool conv) {
  const double speed = conv ? (exact ? c.conv_laser : c.conv_fast) : (exact ? c.laser : c.fast);
  if (speed <= 0.0) return 1e300;
  const int R = exact ? c.r_laser : c.r_fast;
  const int64_t slots = 256 * (int64_t)R;
  const int64_t q = tiles / slots, rem = tiles % slots;
  const int w = (int)((rem + 255) / 256);
  const double tail = w ? (double)w * c.occ[R - 1] / c.occ[w - 1] : 0.0;
  return ((double)(q * R) + tail) * c.bm * c.bn / speed;
}
static int heuristic_cfg(const GemmArgs<float> &a, bool exact, bool need_gen = false, bool conv = false, double *t_out = nullptr) {
  int best = kCfgSmall;
  double best_t = 1e300;
  for (const Cand &c : kCands) {
    if (need_gen && !c.gen) continue;
    const double t = cand_time(c, tiles_of(a, c.bm, c.bn), exact, conv);
    if (t < best_t * 0.999) {  // ties go to the earlier (larger-tile) candidate: less L2 traffic
      best_t = t;
      best = c.cfg;
    }
  }
  if (t_out) *t_out = best_t;
  return best;
}

// Main + tail cut.  The round model above says where a single tile configuration loses: the last, under-filled round
// (C4's 1600 tiles of 128x128 on 512 resident slots are 3.125 rounds and cost 3.6).  Output tiles are independent
// and every configuration is bit-identical, so the columns are cut at n_cut (a multiple of the main tile's BN):
// the main launch covers [0, n_cut) in (nearly) whole rounds of large tiles, the tail launch [n_cut, N) with a
// smaller tile.  Taken only when the model predicts >= 5 % (boundary between the launches ~3 us priced in).
struct SplitPlan {
  int cfg_main = -1, cfg_tail = -1;
  int64_t n_cut = 0;  // 0: one launch
};
std::atomic<int> g_split_tail{1};  // option "split_tail": 0 = never cut, 1 = main launch + tail launch
std::atomic<int64_t> g_last_split{0};  // diagnostics: column cut of the last MFMA GEMM / conv launch (0: single launch)
static SplitPlan plan_split(const GemmArgs<float> &a, bool exact, bool need_gen, bool conv, bool bn_multiple_only) {
  SplitPlan p;
  double t_single;
  p.cfg_main = heuristic_cfg(a, exact, need_gen, conv, &t_single);
  if (!g_split_tail) return p;
  const double boundary = 3.0e6 / (512.0 * (double)a.K);  // ~3 us in model units
  double best = t_single * 0.95;
  for (const Cand &c : kCands) {
    if (need_gen && !c.gen) continue;
    const int64_t tm = (a.M + c.bm - 1) / c.bm, tn = (a.N + c.bn - 1) / c.bn;
    if (tm * tn * a.batch < 256) continue;  // under one round: other mechanisms (slice-parallel, small tiles) apply
    for (int64_t k = tn - 1; k >= 1 && k >= tn - 16; k--) {
      const double t_main = cand_time(c, tm * k * a.batch, exact, conv);
      if (t_main >= best) continue;
      const int64_t n_tail = a.N - k * c.bn;
      for (const Cand &d : kCands) {
        if ((need_gen && !d.gen) || d.bm * d.bn >= c.bm * c.bn) continue;
        if (bn_multiple_only && (k * c.bn) % d.bn != 0) continue;
        const int64_t tt = ((a.M + d.bm - 1) / d.bm) * ((n_tail + d.bn - 1) / d.bn) * a.batch;
        const double t = t_main + cand_time(d, tt, exact, conv) + boundary;
        if (t < best) {
          best = t;
          p.cfg_main = c.cfg;
          p.cfg_tail = d.cfg;
          p.n_cut = k * c.bn;
        }
      }
    }
  }
  return p;
}
static int fallback_exact_cfg(const GemmArgs<float> &) { return kCfgWideExact; }
static int gen_cfg(const GemmArgs<float> &a, bool exact) { return heuristic_cfg(a, exact, true); }


// fp64: three configurations (rule from profiles/r03/f64_cfg_probe.jsonl).
static int heuristic_cfg(const GemmArgs<double> &a, bool, bool = false) {
  if (tiles_of(a, 128, 128) >= 180) return 0;   // (1536^3 = 144 tiles: 44.6 on 64x64 vs 39.9; 1920^3 = 225: 56.0 vs 48.8)
  return tiles_of(a, 64, 64) <= 256 ? 2 : 1;
}
static int fallback_exact_cfg(const GemmArgs<double> &) { return 0; }
static int gen_cfg(const GemmArgs<double> &, bool) { return 1; }

std::atomic<int> g_conv_patch{1}; // implicit conv: B from an LDS input patch where it fits (0: always the per-element gather)
std::atomic<int> g_conv_kslice{1}; // laser-order conv: tail launch as parallel kc slices + ordered combine (0: one workgroup per tail tile)
std::atomic<int> g_last_conv_tail{0};   // diagnostics: how the last convolution's pixel tail ran (0 none, 1 direct kernel, 2 kc slices + combine, 3 one compiler-kernel launch)
std::atomic<int> g_conv_walk{1};         // option "conv_walk": gemm_f32_asm.cpp launch_conv_f32_asm
std::atomic<int> g_conv_cut_always{0};   // option "conv_cut_always" (tests, probes): cut a 3x3 convolution at its last whole 128-pixel tile whatever the model says
std::atomic<int> g_conv_tail{1};   // option "conv_tail": the direct tail kernel behind the assembly main launch (conv_tail.hip); 0 = the round-3 forms
std::atomic<int> g_last_f32_cfg{-1}; // last configuration launch_mfma<float> / the conv launcher ran (diagnostics, tests)

// Main launch, then the tail launch on the same stream.  (Running the tail BESIDE the main launch on a side stream was
// measured worse on every shape tried -- C4 conv 0.590 vs 0.553 ms, 5000^3 2.50 vs 2.24 ms, profiles/r02/conv_c4_v3.log:
// the tail's workgroups take LDS / register slots that delay whole main-launch workgroups -- and is not built.)
template <typename MainFn, typename TailFn>
static hipError_t launch_main_and_tail(hipStream_t s, MainFn &&main_fn, TailFn &&tail_fn) {
  const hipError_t e = main_fn(s);
  return e != hipSuccess ? e : tail_fn(s);
}

// one launch of configuration `cfg` (falling back to a configuration with the scalar loaders when the operands need them)
template <typename E>
static hipError_t launch_mfma_cfg(const GemmArgs<E> &a, const CfgInfo<E> *cfgs, int cfg, bool exact, hipStream_t s) {
  for (int attempt = 0; attempt < 2; attempt++) {
    const CfgInfo<E> &c = cfgs[cfg];
    if (std::is_same<E, float>::value && a.col0 == 0) g_last_f32_cfg = cfg;
    bool va, vb, ea, eb;
    const int am = pick_mode<E>(a.A, a.rsA, a.csA, a.bsA, a.Mext, a.Kext, c.bm, c.bk, &va, &ea);
    const int bm = pick_mode<E>(a.B, a.csB, a.rsB, a.bsB, a.Next, a.Kext, c.bn, c.bk, &vb, &eb);
    if (c.vec && va && vb) return c.fn(a, am, bm, exact, s);
    if (c.vec && (va || ea) && (vb || eb)) return c.fn(a, to_edge(am), to_edge(bm), exact, s);
    if (c.gen) return c.fn(a, to_gen(am), to_gen(bm), exact, s);
    cfg = gen_cfg(a, exact);  // a configuration that carries the scalar (any-stride) loaders
  }
  return hipErrorInvalidValue;
}

static SplitPlan plan_for(const GemmArgs<float> &a, bool exact) { return plan_split(a, exact, false, false, false); }
static SplitPlan plan_for(const GemmArgs<double> &a, bool exact) {
  SplitPlan p;
  p.cfg_main = heuristic_cfg(a, exact);
  return p;
}

template <typename E>
static hipError_t launch_mfma(const GemmArgs<E> &args, const CfgInfo<E> *cfgs, int ncfg, int cfg, bool laser_order,
                              int kc_elems, hipStream_t s) {
  if (args.M <= 0 || args.N <= 0 || args.K <= 0 || args.batch <= 0) return hipSuccess;
  GemmArgs<E> a = args;
  const bool plain_n = a.Next <= a.N;  // not a tile-padded pre-pack image: the readable extent is N itself
  if (a.Mext < a.M) a.Mext = a.M;
  if (a.Next < a.N) a.Next = a.N;
  if (a.Kext < a.K) a.Kext = a.K;
  a.dbg = 0;
  a.col0 = 0;
  // K <= kc is ONE accumulation slice: the single-chain kernel already is Laser's arithmetic, so the
  // second accumulator set of the laser-order kernels is only paid for when K > kc.
  const bool exact = laser_order && a.K > kc_elems;
  a.kc = exact ? kc_elems : 0;  // gemm_tiling.nim:310: kc = 2048 / sizeof(T)
  SplitPlan plan;
  if (cfg < 0 || cfg >= ncfg)
    plan = plan_for(a, exact);
  else
    plan.cfg_main = cfg;
  if (exact && !cfgs[plan.cfg_main].exact) plan.cfg_main = fallback_exact_cfg(a);
  g_last_split = plan.n_cut;
  if (plan.n_cut <= 0) return launch_mfma_cfg<E>(a, cfgs, plan.cfg_main, exact, s);
  GemmArgs<E> m = a;  // columns [0, n_cut): whole tiles of the main configuration
  m.N = plan.n_cut;
  if (plain_n) m.Next = plan.n_cut;
  GemmArgs<E> t = a;  // columns [n_cut, N)
  t.col0 = plan.n_cut;
  return launch_main_and_tail(
      s, [&](hipStream_t q) { return launch_mfma_cfg<E>(m, cfgs, plan.cfg_main, exact, q); },
      [&](hipStream_t q) { return launch_mfma_cfg<E>(t, cfgs, plan.cfg_tail, exact, q); });
}

hipError_t launch_gemm_f32(const GemmArgs<float> &args, int cfg, bool laser_order, hipStream_t s) {
  return launch_mfma<float>(args, kCfgsF32, LH_F32_NUM_CONFIGS, cfg, laser_order, 512, s);
}

hipError_t launch_gemm_f64(const GemmArgs<double> &args, bool laser_order, hipStream_t s) {
  return launch_mfma<double>(args, kCfgsF64, LH_F64_NUM_CONFIGS, -1, laser_order, 256, s);
}

// one launch of the implicit-GEMM convolution with configuration `cfg`
static hipError_t launch_conv_cfg(const GemmArgs<float> &a, int cfg, bool exact, hipStream_t s) {
  for (int attempt = 0; attempt < 2; attempt++) {
    const CfgInfo<float> &c = kCfgsF32[cfg];
    if (a.col0 == 0) g_last_f32_cfg = cfg;
    bool va, ea;
    pick_mode<float>(a.A, a.rsA, a.csA, a.bsA, a.M, a.K, c.bm, c.bk, &va, &ea);
    // B through an LDS-resident input patch when it fits the B region of a stage (else the per-element gather)
    ConvPatchGeom pg;
    // (patch vs gather on the 8-wave 256x128x16 laser-order kernel is box-to-box noise: C4 0.557 vs 0.541 ms in
    // profiles/r02/conv_c4_v6.log, 0.528 vs 0.530 ms in conv_c4_v7.log -- no per-configuration rule)
    const bool patch_ok = g_conv_patch && a.cW % 4 == 0 && a.cpW <= 4 &&
                          conv_patch_geom(c.bk, c.bn, a.cW, a.coW, a.ckH, a.ckW, a.csH, a.csW, &pg);
    const int bmode = patch_ok ? LOAD_CONV_PATCH : LOAD_IM2COL;
    if (a.csA == 1 && va) return c.fn(a, LOAD_VEC_K, bmode, exact, s);
    if (a.csA == 1 && ea) return c.fn(a, LOAD_VEC_K_EDGE, bmode, exact, s);
    if (c.gen) return c.fn(a, LOAD_GEN_K, LOAD_IM2COL, exact, s);
    cfg = heuristic_cfg(a, exact, true, true);
  }
  return hipErrorInvalidValue;
}

// Implicit-GEMM convolution: same kernels, B loader = LOAD_IM2COL / LOAD_CONV_PATCH.  M = C_out, N = oH*oW,
// K = C_in*kH*kW, batch = images; A (the filter bank) is always k-contiguous.
hipError_t launch_conv_implicit_f32(const GemmArgs<float> &args, int cfg, bool laser_order, hipStream_t s) {
  if (args.M <= 0 || args.N <= 0 || args.K <= 0 || args.batch <= 0) return hipSuccess;
  GemmArgs<float> a = args;
  a.Mext = a.M; a.Next = a.N; a.Kext = a.K;
  a.dbg = 0;
  a.col0 = 0;
  a.cs_imgs = 0; a.cs_len = 0;
  g_last_f32_asm = 0;
  const bool exact = laser_order && a.K > 512;
  a.kc = exact ? 512 : 0;
  if (cfg < 0) {  // few output channels, short reduction (the reference's conv bench shape): an HBM stream, not a tile problem
    const hipError_t e = launch_conv_direct_small_f32(a, s);
    if (e != hipErrorNotSupported) {
      g_last_split = 0;
      g_last_f32_cfg = -3;
      return e;
    }
  }
  // the BK=32 laser-order kernel has no registers left for the gather state (it would spill): same tile at BK=16
  auto fix = [&](int c) {
    if (exact && !kCfgsF32[c].exact) c = kCfgWideExact;
    if (exact && c == kCfgWideExact) c = kCfgWide;
    return c;
  };
  SplitPlan plan;
  if (cfg < 0 || cfg >= LH_F32_NUM_CONFIGS)
    plan = plan_split(a, exact, false, true, false);
  else
    plan.cfg_main = cfg;
  // the main part on the hand-scheduled kernels: the cut their 128-pixel tiles want (gemm_f32_asm.cpp conv_asm_plan_cut)
  if (cfg < 0 && g_f32_asm && g_split_tail) {
    const int64_t cut = conv_asm_plan_cut(a, laser_order);
    if (cut >= 0 && cut != plan.n_cut) {
      plan.n_cut = cut;
      if (cut > 0) {
        if (plan.cfg_main < 0 || kCfgsF32[plan.cfg_main].bn > 128) plan.cfg_main = kCfgWide;
        if (plan.cfg_tail < 0) plan.cfg_tail = kCfgSmall;
      }
    }
  }
  if (g_conv_cut_always && cfg < 0 && plan.n_cut == 0 && a.N > 128 && a.N % 128 != 0 && a.ckH == 3 && a.ckW == 3) {
    plan.n_cut = a.N / 128 * 128;           // (the tail forms on shapes the model would leave in one launch)
    plan.cfg_main = kCfgWide;
    plan.cfg_tail = kCfgSmall;
  }
  plan.cfg_main = fix(plan.cfg_main);
  g_last_split = plan.n_cut;
  // the main part (whole 128-pixel tiles, or the whole image) on the hand-scheduled assembly kernel when it is a 3x3 /
  // stride-1 convolution; hipErrorNotSupported = not its class: the compiler-scheduled loaders below
  // (whether the assembly kernel took THIS call's main part is reported through a local: the process-wide diagnostic
  // g_last_f32_asm may be overwritten by another host thread's launch between the launch and a read -- ADVICE r5)
  bool main_on_asm = false;
  const auto main_launch = [&](const GemmArgs<float> &mm, hipStream_t q) {
    if (cfg < 0) {
      const hipError_t e = launch_conv_f32_asm(mm, laser_order, q);
      if (e != hipErrorNotSupported) {
        main_on_asm = e == hipSuccess;
        return e;
      }
    }
    return launch_conv_cfg(mm, plan.cfg_main, exact, q);
  };
  g_last_conv_tail = 0;
  if (plan.n_cut <= 0) return main_launch(a, s);
  GemmArgs<float> m = a;  // output pixels [0, n_cut) of every image
  m.N = plan.n_cut; m.Next = plan.n_cut;
  GemmArgs<float> t = a;  // output pixels [n_cut, oH*oW)
  t.col0 = plan.n_cut;
  const int cfg_tail = fix(plan.cfg_tail);
  // Laser-order tail, K-slice-parallel: the tail launch is a few small workgroups whose K loop is latency-bound (C4: 128
  // workgroups of 64x64, 44 us for 2 % of the work).  Laser's kc slices are independent chains from +0 whose sums are
  // added in order (gemm.nim:150-158), so the tail runs as images x slices workgroup sets, each ONE chain over its 512 k
  // into a workspace, and the ordered combine pass folds them: same fused multiply-adds, same order => bit-identical.
  const int64_t nsl = (a.K + 511) / 512, ntail = a.N - plan.n_cut;
  // (the one-chain mode has no order to keep: it takes the same faster tail)
  // (round 5) behind the hand-scheduled main launch the tail is one launch of the latency-built direct kernel (conv_tail.hip: one
  // wave per 32x32 block and kc slice, operands straight into the MFMA lane layout, ordered fold in LDS -- no workspace, no combine
  // pass); option "conv_tail" = 0: the round-3 forms below
  bool main_done = false;
  if (cfg < 0 && g_conv_tail && g_split_tail) {
    if (hipError_t e = main_launch(m, s); e != hipSuccess) return e;
    main_done = true;
    if (main_on_asm) {
      const hipError_t e = launch_conv_tail_f32(t, 512, s);
      if (e != hipErrorNotSupported) {
        g_last_conv_tail = 1;
        return e;
      }
    }
  }
  if ((exact || !laser_order) && g_conv_kslice && g_split_tail && nsl >= 2 && a.bias == nullptr && a.act == 0 && a.bsC == a.M * a.rsC &&
      (int64_t)a.batch * nsl <= 65535) {
    if (!main_done)
      if (hipError_t e = main_launch(m, s); e != hipSuccess) return e;
    g_last_conv_tail = 2;
    const int64_t mn = a.M * ntail;
    float *W = nullptr;
    if (hipError_t e = scratch_alloc_async((void **)&W, (size_t)(nsl * a.batch * mn) * sizeof(float), s); e != hipSuccess) return e;
    t.alpha = 1.0f; t.beta = 0.0f;
    t.kc = 0;
    t.cs_imgs = a.batch; t.cs_len = 512;
    t.batch = (int32_t)(a.batch * nsl);
    t.C = W - plan.n_cut;  // the kernel addresses C by absolute column: column n_cut + j of slice-image z lands at W[z][i][j]
    t.rsC = ntail; t.csC = 1; t.bsC = mn;
    hipError_t e = launch_conv_cfg(t, plan.cfg_tail, false, s);
    // images stack as rows of one (batch*M) x ntail view of the output's tail columns
    if (e == hipSuccess)
      e = launch_combine_slices<float>(a.C + plan.n_cut * a.csC, a.rsC, a.csC, W, (int64_t)a.batch * a.M, ntail, (int)nsl, a.alpha, a.beta, s);
    const hipError_t e2 = hipFreeAsync(W, s);
    return e != hipSuccess ? e : e2;
  }
  g_last_conv_tail = 3;
  if (main_done) return launch_conv_cfg(t, cfg_tail, exact, s);
  return launch_main_and_tail(
      s, [&](hipStream_t q) { return main_launch(m, q); },
      [&](hipStream_t q) { return launch_conv_cfg(t, cfg_tail, exact, q); });
}

}  // namespace laser_hip
