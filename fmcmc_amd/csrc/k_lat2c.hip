// k_lat2c.hip -- mh_sweep_lat<2, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), the reflective kernels, p = 4 .. 7
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv2c(int p) {
  switch (p) {
    case 4: return (const void*)mh_sweep_lat<2, 4, 10>;
    case 5: return (const void*)mh_sweep_lat<2, 5, 10>;
    case 6: return (const void*)mh_sweep_lat<2, 6, 8>;
    case 7: return (const void*)mh_sweep_lat<2, 7, 8>;
    default: return nullptr;
  }
}
}  // namespace fmh
