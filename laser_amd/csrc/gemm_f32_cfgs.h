// laser_amd/csrc/gemm_f32_cfgs.h -- the tile configurations of the f32 MFMA kernel.
// X(index, BM, BN, BK, WM, WN, WITH_VEC, WITH_GEN, WITH_EXACT)
// Each line is compiled in its own translation unit (gemm_f32_cfg.hip with -DLH_CFG=index).
#pragma once
#define LH_F32_CONFIGS(X)                         \
  X(0, 128, 128, 32, 2, 2, true, true, true)     \
  X(1, 256, 128, 32, 4, 2, true, false, true)    \
  X(2, 128, 128, 16, 2, 2, true, false, true)    \
  X(3, 128, 256, 32, 2, 4, true, false, true)    \
  X(4, 64, 64, 32, 2, 2, true, true, true)       \
  X(5, 256, 256, 16, 2, 4, true, false, false)   \
  X(6, 256, 128, 16, 4, 2, true, false, true)
#define LH_F32_NUM_CONFIGS 7
