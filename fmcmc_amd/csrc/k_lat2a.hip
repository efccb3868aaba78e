// k_lat2a.hip -- mh_sweep_lat<2, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), the reflective kernels, p = 0 .. 2
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv2d();   // k_lat2d.hip: p = 2
FMH_HIDDEN const void* k_lat_kv2a(int p) {
  switch (p) {
    case 0: return (const void*)mh_sweep_lat<2, 0, 20>;     // (iid Normal: the linear model with an intercept and no covariate)
    case 1: return (const void*)mh_sweep_lat<2, 1, 20>;
    case 2: return k_lat_kv2d();
    default: return nullptr;
  }
}
}  // namespace fmh
