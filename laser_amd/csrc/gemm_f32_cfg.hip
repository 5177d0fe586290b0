// laser_amd/csrc/gemm_f32_cfg.hip -- instantiates ONE tile configuration (-DLH_CFG=n) of the f32
// MFMA kernel for every loader-mode pair; one TU per configuration so they compile in parallel.
#include "gemm_f32_cfgs.h"
#include "gemm_f32_mfma_kernel.h"

#ifndef LH_CFG
#error "compile with -DLH_CFG=<configuration index>"
#endif

namespace laser_hip {

template <>
hipError_t launch_gemm_f32_cfg<LH_CFG>(const GemmArgs<float> &a, int amode, int bmode, bool exact,
                                       hipStream_t s) {
  using C = F32Cfg<LH_CFG>;
  if (exact && !C::EXACT) return hipErrorNotSupported;
  if constexpr (C::EXACT) {
    if (exact)
      return launch_cfg_mode<C::BM, C::BN, C::BK, C::WM, C::WN, C::STAGES, C::OCCE, C::VEC, C::GEN, true>(
          a, amode, bmode, s);
  }
  return launch_cfg_mode<C::BM, C::BN, C::BK, C::WM, C::WN, C::STAGES, C::OCCF, C::VEC, C::GEN, false>(
      a, amode, bmode, s);
}

}  // namespace laser_hip
