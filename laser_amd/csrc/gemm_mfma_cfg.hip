// laser_amd/csrc/gemm_mfma_cfg.hip -- instantiates ONE tile configuration (-DLH_CFG=n, float32 list, or
// float64 list with -DLH_F64) of the MFMA GEMM kernel for every loader-mode pair; one TU per
// configuration so they compile in parallel.
#include "gemm_mfma_cfgs.h"
#include "gemm_mfma_kernel.h"

#ifndef LH_CFG
#error "compile with -DLH_CFG=<configuration index>"
#endif

namespace laser_hip {

#ifdef LH_F64
using Elem = double;
using C = F64Cfg<LH_CFG>;
#define LH_ENTRY launch_gemm_f64_cfg<LH_CFG>
#else
using Elem = float;
using C = F32Cfg<LH_CFG>;
#define LH_ENTRY launch_gemm_f32_cfg<LH_CFG>
#endif

template <>
hipError_t LH_ENTRY(const GemmArgs<Elem> &a, int amode, int bmode, bool exact, hipStream_t s) {
  if (exact && !C::EXACT) return hipErrorNotSupported;
  if constexpr (C::EXACT) {
    if (exact)
      return launch_cfg_mode<Elem, C::BM, C::BN, C::BK, C::WM, C::WN, C::STAGES, C::OCCE, C::VEC, C::GEN, true>(
          a, amode, bmode, s);
  }
  return launch_cfg_mode<Elem, C::BM, C::BN, C::BK, C::WM, C::WN, C::STAGES, C::OCCF, C::VEC, C::GEN, false>(
      a, amode, bmode, s);
}

}  // namespace laser_hip
