// scripts/probes/gemm_f32_probe.hip -- TUNING PROBE, built into its OWN library (build/liblaser_probe.so by
// scripts/probes/Makefile); nothing in liblaser_hip.so references it.  The production f32 MFMA kernel template
// with run-time ablation switches (dbg bit 0: skip HBM loads, bit 1: skip LDS stores, bit 2: skip barriers) to
// price each part of the main loop.  Results are WRONG by construction when any switch is on;
// scripts/ablate_f32.py only times it.
#include <cstring>

#include "../../laser_amd/csrc/gemm_mfma_kernel.h"

using namespace laser_hip;

// shape 0: 256x256x16 (2x4 waves of 128x64)   1: 256x128x32 (4x2 waves of 64x64)   2: 256x128x16   3: 128x128x16
// exact != 0: the laser-order form (kc = 512 slice fold, alpha == 1 variant) where the configuration has one
extern "C" int laser_probe_f32(int64_t n, const float *A, const float *B, float *C, int dbg, int shape, int exact,
                               void *stream) {
  GemmArgs<float> g;
  memset(&g, 0, sizeof g);
  g.M = g.N = g.K = n;
  g.alpha = 1.0f; g.beta = 0.0f;
  g.A = A; g.rsA = n; g.csA = 1;
  g.B = B; g.rsB = n; g.csB = 1;
  g.C = C; g.rsC = n; g.csC = 1;
  g.Mext = g.Next = g.Kext = n;
  g.batch = 1;
  g.dbg = dbg & 0xff;
  g.kc = exact ? 512 : 0;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipErrorInvalidValue;
  if (shape == 0 && !exact) e = launch_one<float, 256, 256, 16, 2, 4, LOAD_VEC_K, LOAD_VEC_X, false, 3, 2, true>(g, s);
  if (shape == 1 && !exact) e = launch_one<float, 256, 128, 32, 4, 2, LOAD_VEC_K, LOAD_VEC_X, false, 3, 2, true>(g, s);
  if (shape == 1 && exact) e = launch_one<float, 256, 128, 32, 4, 2, LOAD_VEC_K, LOAD_VEC_X, true, 3, 2, true, false, true>(g, s);
  if (shape == 2 && !exact) e = launch_one<float, 256, 128, 16, 4, 2, LOAD_VEC_K, LOAD_VEC_X, false, 3, 2, true>(g, s);
  if (shape == 2 && exact) e = launch_one<float, 256, 128, 16, 4, 2, LOAD_VEC_K, LOAD_VEC_X, true, 3, 2, true, false, true>(g, s);
  if (shape == 3 && !exact) e = launch_one<float, 128, 128, 16, 2, 2, LOAD_VEC_K, LOAD_VEC_X, false, 3, 3, true>(g, s);
  if (shape == 3 && exact) e = launch_one<float, 128, 128, 16, 2, 2, LOAD_VEC_K, LOAD_VEC_X, true, 3, 2, true, false, true>(g, s);
  return (int)e;
}
