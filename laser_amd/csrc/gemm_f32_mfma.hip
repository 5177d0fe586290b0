// laser_amd/csrc/gemm_f32_mfma.hip -- host-side dispatch for the f32 MFMA GEMM: picks the tile
// configuration and, per operand, the HBM->LDS "packing" loader that matches its strides.
// This replaces the reference's run-time ISA dispatch (gemm.nim:228-247) and its Tiles/partitionMNK
// geometry (gemm_tiling.nim:276-341) -- on the GPU the geometry is the workgroup tile.
#include "common.h"
#include "gemm_f32_cfgs.h"

namespace laser_hip {

#define X(IDX, BM, BN, BK, WM, WN, WV, WG, WE) \
  hipError_t launch_gemm_f32_cfg##IDX(const GemmArgs<float> &, int, int, bool, hipStream_t);
LH_F32_CONFIGS(X)
#undef X

struct CfgInfo {
  int bm, bn, bk, wm, wn;
  bool vec, gen, exact;
  const char *name;
  hipError_t (*fn)(const GemmArgs<float> &, int, int, bool, hipStream_t);
};

#define X(IDX, BM, BN, BK, WM, WN, WV, WG, WE) \
  {BM, BN, BK, WM, WN, WV, WG, WE, #BM "x" #BN "x" #BK "_w" #WM "x" #WN, launch_gemm_f32_cfg##IDX},
static const CfgInfo kCfgs[LH_F32_NUM_CONFIGS] = {LH_F32_CONFIGS(X)};
#undef X

int gemm_f32_config_count() { return LH_F32_NUM_CONFIGS; }
const char *gemm_f32_config_name(int cfg) {
  return (cfg >= 0 && cfg < LH_F32_NUM_CONFIGS) ? kCfgs[cfg].name : "?";
}

static inline int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

// Can operand X (panel along `x` with stride sx, k with stride sk) use 16-B vector loads for this
// tile shape?  Needs: unit stride along one axis, the other stride and the batch stride multiples of
// 4 elements, 16-B aligned base, and no ragged tile in x or k.
static int pick_mode(const float *p, int64_t sx, int64_t sk, int64_t bs, int64_t X, int64_t K, int bx,
                     int bk, bool *vec_ok) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (bs % 4 == 0);
  const bool full = (X % bx == 0) && (K % bk == 0);
  if (sk == 1) {
    *vec_ok = aligned && full && (sx % 4 == 0);
    return LOAD_VEC_K;
  }
  if (sx == 1) {
    *vec_ok = aligned && full && (sk % 4 == 0);
    return LOAD_VEC_X;
  }
  *vec_ok = false;
  return iabs64(sk) <= iabs64(sx) ? LOAD_VEC_K : LOAD_VEC_X;
}

static int to_gen(int mode) { return mode == LOAD_VEC_K ? LOAD_GEN_K : LOAD_GEN_X; }

static int heuristic_cfg(const GemmArgs<float> &a, bool exact) {
  (void)exact;
  // Small problems: more, smaller tiles so the 256 CUs have something to do.
  const int64_t tiles128 = ((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
  if (tiles128 < 128) return 4;
  return 0;
}

hipError_t launch_gemm_f32(const GemmArgs<float> &args, int cfg, bool laser_order, hipStream_t s) {
  if (args.M <= 0 || args.N <= 0 || args.K <= 0 || args.batch <= 0) return hipSuccess;
  GemmArgs<float> a = args;
  if (a.Mext < a.M) a.Mext = a.M;
  if (a.Next < a.N) a.Next = a.N;
  if (a.Kext < a.K) a.Kext = a.K;
  a.kc = laser_order ? 512 : 0;  // gemm_tiling.nim:310: kc = 2048 / sizeof(float32)
  if (cfg < 0 || cfg >= LH_F32_NUM_CONFIGS) cfg = heuristic_cfg(a, laser_order);
  if (laser_order && !kCfgs[cfg].exact) cfg = 1;
  for (int attempt = 0; attempt < 2; attempt++) {
    const CfgInfo &c = kCfgs[cfg];
    bool va, vb;
    int am = pick_mode(a.A, a.rsA, a.csA, a.bsA, a.Mext, a.Kext, c.bm, c.bk, &va);
    int bm = pick_mode(a.B, a.csB, a.rsB, a.bsB, a.Next, a.Kext, c.bn, c.bk, &vb);
    if (va && vb && c.vec) return c.fn(a, am, bm, laser_order, s);
    if (c.gen) return c.fn(a, to_gen(am), to_gen(bm), laser_order, s);
    cfg = (((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch < 128) ? 4 : 0;  // has GEN loaders
  }
  return hipErrorInvalidValue;
}

}  // namespace laser_hip
