// k_lat3b.hip -- mh_sweep_lat<2, P, 4> (mh_lat.hpp): the latency form of the linear model with 8 .. 15 covariates (up to 2048 observations:
// four slots of P + 1 doubles per lane), the reflective kernels
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv2w(int p) {
  switch (p) {
    case 8: return (const void*)mh_sweep_lat<2, 8, 4>;
    case 9: return (const void*)mh_sweep_lat<2, 9, 4>;
    case 10: return (const void*)mh_sweep_lat<2, 10, 4>;
    case 11: return (const void*)mh_sweep_lat<2, 11, 4>;
    case 12: return (const void*)mh_sweep_lat<2, 12, 4>;
    case 13: return (const void*)mh_sweep_lat<2, 13, 4>;
    case 14: return (const void*)mh_sweep_lat<2, 14, 4>;
    case 15: return (const void*)mh_sweep_lat<2, 15, 4>;
    default: return nullptr;
  }
}
}  // namespace fmh
