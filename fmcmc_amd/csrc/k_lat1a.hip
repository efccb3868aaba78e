// k_lat1a.hip -- mh_sweep_lat<1, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), kernel_normal / kernel_unif, p = 0 .. 2
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv1d();   // k_lat1d.hip: p = 2
FMH_HIDDEN const void* k_lat_kv1a(int p) {
  switch (p) {
    case 0: return (const void*)mh_sweep_lat<1, 0, 20>;     // (iid Normal: the linear model with an intercept and no covariate)
    case 1: return (const void*)mh_sweep_lat<1, 1, 20>;
    case 2: return k_lat_kv1d();
    default: return nullptr;
  }
}
}  // namespace fmh
