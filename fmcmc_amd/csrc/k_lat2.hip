// k_lat2.hip -- mh_sweep_lat<2, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), the reflective kernels
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv2(int p) {
  switch (p) {
    case 0: return (const void*)mh_sweep_lat<2, 0, 20>;     // (iid Normal: the linear model with an intercept and no covariate)
    case 1: return (const void*)mh_sweep_lat<2, 1, 20>;
    case 2: return (const void*)mh_sweep_lat<2, 2, 20>;
    case 3: return (const void*)mh_sweep_lat<2, 3, 20>;
    case 4: return (const void*)mh_sweep_lat<2, 4, 10>;
    case 5: return (const void*)mh_sweep_lat<2, 5, 10>;
    case 6: return (const void*)mh_sweep_lat<2, 6, 8>;
    case 7: return (const void*)mh_sweep_lat<2, 7, 8>;
    default: return nullptr;
  }
}
}  // namespace fmh
