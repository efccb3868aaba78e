// k_lat_l2b.hip -- mh_sweep_lat<2, P, OPTMAX, LOGISTIC> (mh_lat.hpp): the latency form for the logistic family (round 5), the reflective kernels, p = 4 .. 7
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_lg2b(int p) {
  switch (p) {
    case 4: return (const void*)mh_sweep_lat<2, 4, 10, FMCMC_FAM_LOGISTIC>;
    case 5: return (const void*)mh_sweep_lat<2, 5, 10, FMCMC_FAM_LOGISTIC>;
    case 6: return (const void*)mh_sweep_lat<2, 6, 8, FMCMC_FAM_LOGISTIC>;
    case 7: return (const void*)mh_sweep_lat<2, 7, 8, FMCMC_FAM_LOGISTIC>;
    default: return nullptr;
  }
}
}  // namespace fmh
