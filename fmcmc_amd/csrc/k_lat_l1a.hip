// k_lat_l1a.hip -- mh_sweep_lat<1, P, OPTMAX, LOGISTIC> (mh_lat.hpp): the latency form for the logistic family (round 5), kernel_normal / kernel_unif, p = 1 .. 3
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_lg1c();   // k_lat_l1c.hip: p = 3
FMH_HIDDEN const void* k_lat_lg1a(int p) {
  switch (p) {
    case 1: return (const void*)mh_sweep_lat<1, 1, 20, FMCMC_FAM_LOGISTIC>;
    case 2: return (const void*)mh_sweep_lat<1, 2, 20, FMCMC_FAM_LOGISTIC>;
    case 3: return k_lat_lg1c();
    default: return nullptr;
  }
}
}  // namespace fmh
