// k_spec_m.hip -- mh_sweep_spec<P, OPTMAX, KIND> (mh_spec.hpp) for the mirror kernels (KIND 7, 8: kernel_nmirror / kernel_umirror,
// R/kernel_mirror.R): their owner (mfma_owner_mirror) on the wave-specialised kernel's LDS sequence words, one to four chains per workgroup
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_mirror(int p, int kind) {
  if (kind != FMCMC_KERNEL_NMIRROR && kind != FMCMC_KERNEL_UMIRROR) return nullptr;
#define SPEC_M(PV, OV) ((kind == FMCMC_KERNEL_NMIRROR) ? (const void*)mh_sweep_spec<PV, OV, FMCMC_KERNEL_NMIRROR> : (const void*)mh_sweep_spec<PV, OV, FMCMC_KERNEL_UMIRROR>)
  switch (p) {
    case 0: return SPEC_M(0, 20);   // (no covariate: the iid Normal family)
    case 1: return SPEC_M(1, 20);
    case 2: return SPEC_M(2, 20);
    case 3: return SPEC_M(3, 20);
    case 4: return SPEC_M(4, 10);
    case 5: return SPEC_M(5, 10);
    case 6: return SPEC_M(6, 8);
    case 7: return SPEC_M(7, 8);
    case 8: return SPEC_M(8, 4);     // (8 .. 14 covariates on up to 2048 observations)
    case 9: return SPEC_M(9, 4);
    case 10: return SPEC_M(10, 4);
    case 11: return SPEC_M(11, 4);
    case 12: return SPEC_M(12, 4);
    case 13: return SPEC_M(13, 4);
    case 14: return SPEC_M(14, 4);
    default: return nullptr;
  }
#undef SPEC_M
}
}  // namespace fmh
