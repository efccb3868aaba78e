// k_lat1c.hip -- mh_sweep_lat<1, P, OPTMAX> (mh_lat.hpp): the latency form (one to three chains per workgroup), kernel_normal / kernel_unif, p = 4 .. 7
#include "mh_tu.hpp"
#include "mh_lat.hpp"

namespace fmh {
FMH_HIDDEN const void* k_lat_kv1c(int p) {
  switch (p) {
    case 4: return (const void*)mh_sweep_lat<1, 4, 10>;
    case 5: return (const void*)mh_sweep_lat<1, 5, 10>;
    case 6: return (const void*)mh_sweep_lat<1, 6, 8>;
    case 7: return (const void*)mh_sweep_lat<1, 7, 8>;
    default: return nullptr;
  }
}
FMH_HIDDEN const void* k_lat_kv1a(int p);
FMH_HIDDEN const void* k_lat_kv1b(int p);
FMH_HIDDEN const void* k_lat_kv2a(int p);
FMH_HIDDEN const void* k_lat_kv2b(int p);
FMH_HIDDEN const void* k_lat_kv2c(int p);
FMH_HIDDEN const void* k_lat_kv1w(int p);   // k_lat3a.hip / k_lat3b.hip: p = 8 .. 15
FMH_HIDDEN const void* k_lat_kv2w(int p);
const void* k_lat(int p, int kind) {
  if (kind == FMCMC_KERNEL_NORMAL) return p <= 2 ? k_lat_kv1a(p) : (p == 3 ? k_lat_kv1b(p) : (p <= 7 ? k_lat_kv1c(p) : k_lat_kv1w(p)));
  if (kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) return p <= 2 ? k_lat_kv2a(p) : (p == 3 ? k_lat_kv2b(p) : (p <= 7 ? k_lat_kv2c(p) : k_lat_kv2w(p)));
  return nullptr;
}
}  // namespace fmh
