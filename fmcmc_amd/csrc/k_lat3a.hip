// k_lat3a.hip -- mh_sweep_lat<1, P, 4> (mh_lat.hpp): the latency form of the linear model with 8 .. 15 covariates (up to 2048 observations:
// four slots of P + 1 doubles per lane), kernel_normal / kernel_unif
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv1w(int p) {
  switch (p) {
    case 8: return (const void*)mh_sweep_lat<1, 8, 4>;
    case 9: return (const void*)mh_sweep_lat<1, 9, 4>;
    case 10: return (const void*)mh_sweep_lat<1, 10, 4>;
    case 11: return (const void*)mh_sweep_lat<1, 11, 4>;
    case 12: return (const void*)mh_sweep_lat<1, 12, 4>;
    case 13: return (const void*)mh_sweep_lat<1, 13, 4>;
    case 14: return (const void*)mh_sweep_lat<1, 14, 4>;
    case 15: return (const void*)mh_sweep_lat<1, 15, 4>;
    default: return nullptr;
  }
}
}  // namespace fmh
