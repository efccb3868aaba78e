// k_spec_lw1.hip -- mh_sweep_spec<P, 4, 3, LOGISTIC> (mh_spec.hpp): kernel_adapt of the logistic family with 8 .. 15 covariates on up to 2048
// observations (four slots of P doubles per compute lane), the register owner at the compile-time width k <= 16
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_lw1(int p) {
  switch (p) {
    case 8: return (const void*)mh_sweep_spec<8, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 9: return (const void*)mh_sweep_spec<9, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 10: return (const void*)mh_sweep_spec<10, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 11: return (const void*)mh_sweep_spec<11, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 12: return (const void*)mh_sweep_spec<12, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 13: return (const void*)mh_sweep_spec<13, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 14: return (const void*)mh_sweep_spec<14, 4, 3, FMCMC_FAM_LOGISTIC>;
    case 15: return (const void*)mh_sweep_spec<15, 4, 3, FMCMC_FAM_LOGISTIC>;
    default: return nullptr;
  }
}
}  // namespace fmh
