// laser_amd/csrc/gemm_f32_cfg.hip -- instantiates ONE tile configuration (-DLH_CFG=n) of the f32
// MFMA kernel for every loader-mode pair; one TU per configuration so they compile in parallel.
#include "gemm_f32_cfgs.h"
#include "gemm_f32_mfma_kernel.h"

namespace laser_hip {

template <int BM, int BN, int BK, int WM, int WN, bool WV, bool WG, bool WE>
static hipError_t run_cfg(const GemmArgs<float> &a, int amode, int bmode, bool exact, hipStream_t s) {
  if (exact && !WE) return hipErrorNotSupported;
  if constexpr (WE) {
    if (exact) return launch_cfg_mode<BM, BN, BK, WM, WN, WV, WG, true>(a, amode, bmode, s);
  }
  return launch_cfg_mode<BM, BN, BK, WM, WN, WV, WG, false>(a, amode, bmode, s);
}

#define LH_CAT2(a, b) a##b
#define LH_CAT(a, b) LH_CAT2(a, b)
#define X(IDX, BM, BN, BK, WM, WN, WV, WG, WE)                                                      \
  LH_IF_##IDX(hipError_t LH_CAT(launch_gemm_f32_cfg, IDX)(const GemmArgs<float> &a, int amode,      \
                                                         int bmode, bool exact, hipStream_t s) {    \
    return run_cfg<BM, BN, BK, WM, WN, WV, WG, WE>(a, amode, bmode, exact, s);                      \
  })
#define LH_EMPTY(...)
#define LH_KEEP(...) __VA_ARGS__
#if LH_CFG == 0
#define LH_IF_0 LH_KEEP
#else
#define LH_IF_0 LH_EMPTY
#endif
#if LH_CFG == 1
#define LH_IF_1 LH_KEEP
#else
#define LH_IF_1 LH_EMPTY
#endif
#if LH_CFG == 2
#define LH_IF_2 LH_KEEP
#else
#define LH_IF_2 LH_EMPTY
#endif
#if LH_CFG == 3
#define LH_IF_3 LH_KEEP
#else
#define LH_IF_3 LH_EMPTY
#endif
#if LH_CFG == 4
#define LH_IF_4 LH_KEEP
#else
#define LH_IF_4 LH_EMPTY
#endif
#if LH_CFG == 5
#define LH_IF_5 LH_KEEP
#else
#define LH_IF_5 LH_EMPTY
#endif
#if LH_CFG == 6
#define LH_IF_6 LH_KEEP
#else
#define LH_IF_6 LH_EMPTY
#endif
LH_F32_CONFIGS(X)
#undef X

}  // namespace laser_hip
