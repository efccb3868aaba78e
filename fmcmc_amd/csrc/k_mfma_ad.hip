// k_mfma_ad.hip -- mh_sweep_mfma_ad<KIND, NG, KX, BND, NSV> (mh_mfma_ad.hpp): streamed MFMA evaluation with the adaptive owners
// (kernel_adapt, kernel_ram incl. its bounded form, the mirror kernels) between barriers
#include "mh_tu.hpp"
#include "mh_mfma.hpp"
#include "mh_spec.hpp"
#include "mh_mfma_ad.hpp"

namespace fmh {
const void* k_mfma_ad(int kind, int ng, int kx, int bnd, int shrt) {
  // register-row owners: (ng, kx) = (1, 5), (1, 0), (2, 9), (2, 0)
#define AD_REG(GV, XV)                                                                                                            \
  if (ng == GV && kx == XV) {                                                                                                     \
    if (kind == FMCMC_KERNEL_ADAPT && !bnd) {                                                                                     \
      if (shrt) { if constexpr (GV == 1 && XV == 0) return (const void*)mh_sweep_mfma_ad<3, 1, 0, false, 1>; else return nullptr; } \
      return (const void*)mh_sweep_mfma_ad<3, GV, XV>;                                                                            \
    }                                                                                                                             \
    if (kind == FMCMC_KERNEL_RAM && !bnd) {                                                                                       \
      if (shrt) { if constexpr (GV == 1 && XV == 0) return (const void*)mh_sweep_mfma_ad<4, 1, 0, false, 1>; else return nullptr; } \
      return (const void*)mh_sweep_mfma_ad<4, GV, XV>;                                                                            \
    }                                                                                                                             \
    if (kind == FMCMC_KERNEL_RAM) return shrt ? (const void*)mh_sweep_mfma_ad<4, GV, XV, true, 1> : (const void*)mh_sweep_mfma_ad<4, GV, XV, true>; \
    return nullptr;                                                                                                               \
  }
  AD_REG(1, 5) AD_REG(1, 0) AD_REG(2, 9) AD_REG(2, 0)
#undef AD_REG
  // matrices in LDS (kx = -1): 8 .. 15 covariates, or a fixed parameter;  mirror kernels (kx = -2)
#define AD_LDS(GV)                                                                                                                \
  if (ng == GV && kx == -1 && !bnd) {                                                                                             \
    if (kind == FMCMC_KERNEL_ADAPT) return shrt ? (const void*)mh_sweep_mfma_ad<3, GV, -1, false, 1> : (const void*)mh_sweep_mfma_ad<3, GV, -1>; \
    if (kind == FMCMC_KERNEL_RAM) return shrt ? (const void*)mh_sweep_mfma_ad<4, GV, -1, false, 1> : (const void*)mh_sweep_mfma_ad<4, GV, -1>;   \
    return nullptr;                                                                                                               \
  }                                                                                                                               \
  if (ng == GV && kx == -2 && !bnd) {                                                                                             \
    if (kind == FMCMC_KERNEL_NMIRROR) return shrt ? (const void*)mh_sweep_mfma_ad<FMCMC_KERNEL_NMIRROR, GV, -2, false, 1> : (const void*)mh_sweep_mfma_ad<FMCMC_KERNEL_NMIRROR, GV, -2>; \
    if (kind == FMCMC_KERNEL_UMIRROR) return shrt ? (const void*)mh_sweep_mfma_ad<FMCMC_KERNEL_UMIRROR, GV, -2, false, 1> : (const void*)mh_sweep_mfma_ad<FMCMC_KERNEL_UMIRROR, GV, -2>; \
    return nullptr;                                                                                                               \
  }
  AD_LDS(1) AD_LDS(2) AD_LDS(3) AD_LDS(4)
#undef AD_LDS
  return nullptr;
}
}  // namespace fmh
