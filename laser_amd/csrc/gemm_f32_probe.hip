// laser_amd/csrc/gemm_f32_probe.hip -- TUNING PROBE ONLY (not reachable from the production dispatch):
// the 256x256x16 3-stage fast kernel with run-time ablation switches (dbg bit 0: skip HBM loads,
// bit 1: skip LDS stores, bit 2: skip barriers) to price each part of the main loop.  Results are
// WRONG by construction when any switch is on; scripts/ablate_f32.py only times it.
#include "gemm_mfma_kernel.h"

namespace laser_hip {
hipError_t launch_gemm_f32_probe(const GemmArgs<float> &a, int dbg, hipStream_t s) {
  GemmArgs<float> g = a;
  g.dbg = dbg & 0xff;
  g.kc = 0;
  g.Mext = g.M; g.Next = g.N; g.Kext = g.K;
  const int shape = dbg >> 8;  // 0: 256x256x16 (2x4 waves of 128x64)   1: 256x128x32 (4x2 waves of 64x64)   2: 256x128x16
  if (shape == 1) return launch_one<float, 256, 128, 32, 4, 2, LOAD_VEC_K, LOAD_VEC_X, false, 3, 2, true>(g, s);
  if (shape == 2) return launch_one<float, 256, 128, 16, 4, 2, LOAD_VEC_K, LOAD_VEC_X, false, 3, 2, true>(g, s);
  return launch_one<float, 256, 256, 16, 2, 4, LOAD_VEC_K, LOAD_VEC_X, false, 3, 2, true>(g, s);
}
}  // namespace laser_hip
