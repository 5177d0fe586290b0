// laser_amd/csrc/gemm_mfma_cfgs.h -- the tile configurations of the f32 / f64 MFMA kernels.
// X(index, BM, BN, BK, WM, WN, STAGES, OCC_FAST, OCC_EXACT, WITH_VEC, WITH_GEN, WITH_EXACT)
//   BM x BN x BK   workgroup tile;  WM x WN  waves (wave tile = BM/WM x BN/WN, built of 32x32 MFMAs)
//   STAGES         LDS stages (2: barrier per tile end; 3: ring with mid-tile barrier)
//   OCC_*          waves per SIMD requested from the register allocator (fast / laser-order kernels)
//   WITH_GEN       also build the predicated scalar loaders (arbitrary strides, ragged edges)
//   WITH_EXACT     also build the laser-order kernel (needs a second accumulator set in registers)
// Each line is compiled in its own translation unit (gemm_mfma_cfg.hip with -DLH_CFG=index, plus
// -DLH_F64 for the float64 list).
#pragma once
#include "common.h"
#define LH_F32_CONFIGS(X)                                        \
  X(0, 256, 256, 16, 2, 4, 3, 2, 2, true, false, false)          \
  X(1, 256, 128, 16, 4, 2, 3, 2, 2, true, false, true)           \
  X(2, 128, 128, 16, 2, 2, 3, 3, 2, true, true, true)            \
  X(3, 64, 64, 32, 2, 2, 2, 3, 2, true, true, true)              \
  X(4, 256, 128, 32, 4, 2, 3, 2, 2, true, false, true)
#define LH_F32_NUM_CONFIGS 5
// Measured on MI355X at 8192^3 (profiles/r01/sweep_f32_v6.json): cfg 0 fast 138-140 TFLOP/s;
// laser-order (second accumulator set => 1 workgroup/CU on every 64x64-wave tile): cfg 4 129.6-131.3,
// cfg 2 127.4, cfg 1 126.2; fast cfg 1 (2 workgroups/CU) 134.8, cfg 2 133.3, cfg 3 118-122.
// Rejected by measurement (kept out of the build): 2-stage forms of the same tiles (-3..-8 %),
// 128x256x32 (= cfg 4 within noise), 4-wave 256x128 / 128x256 tiles at 1 wave/SIMD for laser-order
// (-8 %), fragment prefetch distance 2 (no gain at 1 WG/CU, costs the 128-VGPR step of cfg 1 fast).
// Mid sizes (1024^3..4100^3, scripts/heuristic_check.py with three extra configurations built in): a
// 3-stage 64x64x32 (+4 % at 1024^3 only), 128x64x16 (never best) and an 8-wave 128x128x16 (+8 % at 2048^3
// laser-order only, where the grid is exactly 256 tiles) -- not worth three more translation units.
// One wave per SIMD (4-wave workgroups, 512 registers per wave, accumulators in AGPRs) -- the shape the vendor library's
// assembly kernels use (torch.matmul -> rocBLAS / hipBLASLt on the same box: 154 TFLOP/s at 8192^3,
// profiles/r02/vendor_blas_v1.jsonl) -- measured again in round 2 with the k-quad image: 256x256x16 as 2x2 waves of
// 128x128 (16 MFMA blocks per wave, 512 registers, no spills) runs 134.1 TFLOP/s fast against 139.7 for the 8-wave form of
// the same tile: without a partner wave every barrier and every counted wait idles the matrix pipe, and the compiler's
// schedule does not close that gap.  The laser-order sibling (256x128x32 as 2x2 waves of 128x64, acc + run = 256
// registers) spills 120-286 VGPRs as written: VALU cannot read AGPRs, so the running sum lands in VGPRs next to the
// fragment and staging registers.  Neither is built.
// cfg 3 as a 3-stage ring with the k-quad image: +3 % at 1024^3, -4 % at 2048^3, -3 % at 8192^3: stays 2-stage.

// float64 (v_mfma_f64_16x16x4_f64, 16x16 blocks, 4 k per instruction): the 8-wave 128x128 tile has the
// same cadence as f32 cfg 0 (8 MFMAs of 64 cycles per k-step); laser-order needs acc 64 + run 64 regs.
// cfg 2 (64x32): problems of at most one round of 64x64 tiles -- 960^3 (the reference's f64 bench shape,
// benchmarks/gemm/gemm_bench_float64.nim) 37.0 / 39.0 vs 35.3 / 35.7 TFLOP/s, 1024^3 42.0 / 43.9 vs 40.4 / 40.6
// (profiles/r03/f64_cfg_probe.jsonl; a 32x32 tile was no better and is not built).
#define LH_F64_CONFIGS(X)                                        \
  X(0, 128, 128, 16, 2, 4, 3, 2, 2, true, false, true)           \
  X(1, 64, 64, 16, 2, 2, 2, 2, 2, true, true, true)              \
  X(2, 64, 32, 16, 2, 2, 2, 2, 2, true, true, true)
#define LH_F64_NUM_CONFIGS 3

namespace laser_hip {
template <int IDX>
struct F32Cfg;
template <int IDX>
struct F64Cfg;
#define X(IDX, BM_, BN_, BK_, WM_, WN_, ST_, OF_, OE_, WV_, WG_, WE_)                                  \
  template <>                                                                                          \
  struct F32Cfg<IDX> {                                                                                 \
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, STAGES = ST_, OCCF = OF_,   \
                         OCCE = OE_;                                                                   \
    static constexpr bool VEC = WV_, GEN = WG_, EXACT = WE_;                                           \
  };
LH_F32_CONFIGS(X)
#undef X
#define X(IDX, BM_, BN_, BK_, WM_, WN_, ST_, OF_, OE_, WV_, WG_, WE_)                                  \
  template <>                                                                                          \
  struct F64Cfg<IDX> {                                                                                 \
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, STAGES = ST_, OCCF = OF_,   \
                         OCCE = OE_;                                                                   \
    static constexpr bool VEC = WV_, GEN = WG_, EXACT = WE_;                                           \
  };
LH_F64_CONFIGS(X)
#undef X

// defined in gemm_mfma_cfg.hip (one explicit specialisation per translation unit)
template <int IDX>
hipError_t launch_gemm_f32_cfg(const GemmArgs<float> &a, int amode, int bmode, bool exact, hipStream_t s);
#define X(IDX, ...) \
  template <>       \
  hipError_t launch_gemm_f32_cfg<IDX>(const GemmArgs<float> &, int, int, bool, hipStream_t);
LH_F32_CONFIGS(X)
#undef X
template <int IDX>
hipError_t launch_gemm_f64_cfg(const GemmArgs<double> &a, int amode, int bmode, bool exact, hipStream_t s);
#define X(IDX, ...) \
  template <>       \
  hipError_t launch_gemm_f64_cfg<IDX>(const GemmArgs<double> &, int, int, bool, hipStream_t);
LH_F64_CONFIGS(X)
#undef X
}  // namespace laser_hip
