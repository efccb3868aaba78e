// k_lat2d.hip -- mh_sweep_lat<2, 2, 20> (mh_lat.hpp): the latency form, p = 2 (a unit of its own: the 20-slot instantiations are the
// longest compiles of the library)
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv2d() { return (const void*)mh_sweep_lat<2, 2, 20>; }
}  // namespace fmh
