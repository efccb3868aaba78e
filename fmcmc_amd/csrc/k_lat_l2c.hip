// k_lat_l2c.hip -- mh_sweep_lat<2, 3, 20, LOGISTIC> (mh_lat.hpp): the latency form for the logistic family, p = 3 (a unit of its own: the
// 20-slot instantiations are the longest compiles of the library)
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_lg2c() { return (const void*)mh_sweep_lat<2, 3, 20, FMCMC_FAM_LOGISTIC>; }
}  // namespace fmh
