// k_lat_l3b.hip -- mh_sweep_lat<2, P, 4, LOGISTIC> (mh_lat.hpp): the latency form for the logistic family with 8 .. 15 covariates (up to
// 2048 observations: four slots of P doubles per lane), the reflective kernels
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_lg2w(int p) {
  switch (p) {
    case 8: return (const void*)mh_sweep_lat<2, 8, 4, FMCMC_FAM_LOGISTIC>;
    case 9: return (const void*)mh_sweep_lat<2, 9, 4, FMCMC_FAM_LOGISTIC>;
    case 10: return (const void*)mh_sweep_lat<2, 10, 4, FMCMC_FAM_LOGISTIC>;
    case 11: return (const void*)mh_sweep_lat<2, 11, 4, FMCMC_FAM_LOGISTIC>;
    case 12: return (const void*)mh_sweep_lat<2, 12, 4, FMCMC_FAM_LOGISTIC>;
    case 13: return (const void*)mh_sweep_lat<2, 13, 4, FMCMC_FAM_LOGISTIC>;
    case 14: return (const void*)mh_sweep_lat<2, 14, 4, FMCMC_FAM_LOGISTIC>;
    case 15: return (const void*)mh_sweep_lat<2, 15, 4, FMCMC_FAM_LOGISTIC>;
    default: return nullptr;
  }
}
}  // namespace fmh
