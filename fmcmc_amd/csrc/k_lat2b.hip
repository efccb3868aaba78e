// k_lat2b.hip -- mh_sweep_lat<2, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), the reflective kernels, p = 3 (C2's shape)
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv2b(int p) {
  switch (p) {
    case 3: return (const void*)mh_sweep_lat<2, 3, 20>;
    default: return nullptr;
  }
}
}  // namespace fmh
