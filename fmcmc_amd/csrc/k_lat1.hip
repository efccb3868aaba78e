// k_lat1.hip -- mh_sweep_lat<1, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), kernel_normal / kernel_unif
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv1(int p) {
  switch (p) {
    case 0: return (const void*)mh_sweep_lat<1, 0, 20>;     // (iid Normal: the linear model with an intercept and no covariate)
    case 1: return (const void*)mh_sweep_lat<1, 1, 20>;
    case 2: return (const void*)mh_sweep_lat<1, 2, 20>;
    case 3: return (const void*)mh_sweep_lat<1, 3, 20>;
    case 4: return (const void*)mh_sweep_lat<1, 4, 10>;
    case 5: return (const void*)mh_sweep_lat<1, 5, 10>;
    case 6: return (const void*)mh_sweep_lat<1, 6, 8>;
    case 7: return (const void*)mh_sweep_lat<1, 7, 8>;
    default: return nullptr;
  }
}
FMH_HIDDEN const void* k_lat_kv2(int p);
const void* k_lat(int p, int kind) { return kind == FMCMC_KERNEL_NORMAL ? k_lat_kv1(p) : kind == FMCMC_KERNEL_NORMAL_REFLECTIVE ? k_lat_kv2(p) : nullptr; }
}  // namespace fmh
