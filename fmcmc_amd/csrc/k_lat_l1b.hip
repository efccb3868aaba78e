// k_lat_l1b.hip -- mh_sweep_lat<1, P, OPTMAX, LOGISTIC> (mh_lat.hpp): the latency form for the logistic family (round 5), kernel_normal / kernel_unif, p = 4 .. 7
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_lg1b(int p) {
  switch (p) {
    case 4: return (const void*)mh_sweep_lat<1, 4, 10, FMCMC_FAM_LOGISTIC>;
    case 5: return (const void*)mh_sweep_lat<1, 5, 10, FMCMC_FAM_LOGISTIC>;
    case 6: return (const void*)mh_sweep_lat<1, 6, 8, FMCMC_FAM_LOGISTIC>;
    case 7: return (const void*)mh_sweep_lat<1, 7, 8, FMCMC_FAM_LOGISTIC>;
    default: return nullptr;
  }
}
FMH_HIDDEN const void* k_lat_lg1a(int p);
FMH_HIDDEN const void* k_lat_lg2a(int p);
FMH_HIDDEN const void* k_lat_lg2b(int p);
FMH_HIDDEN const void* k_lat_lg1w(int p);   // k_lat_l3a.hip / k_lat_l3b.hip: p = 8 .. 15
FMH_HIDDEN const void* k_lat_lg2w(int p);
FMH_HIDDEN const void* k_lat_logit(int p, int kind) {
  if (kind == FMCMC_KERNEL_NORMAL) return p <= 3 ? k_lat_lg1a(p) : (p <= 7 ? k_lat_lg1b(p) : k_lat_lg1w(p));
  if (kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) return p <= 3 ? k_lat_lg2a(p) : (p <= 7 ? k_lat_lg2b(p) : k_lat_lg2w(p));
  return nullptr;
}
FMH_HIDDEN size_t k_lat_logit_lds() { return lat_logit_lds_bytes(); }
}  // namespace fmh
