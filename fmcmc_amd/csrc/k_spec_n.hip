// k_spec_n.hip -- mh_sweep_spec<P, OPTMAX, KIND> (mh_spec.hpp) for the normal / uniform kernels (KIND 1, 2): the LATENCY form
// of few chains per GPU (one to three chains per workgroup); with four chains per workgroup mh_sweep_mfma is the kernel
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
int k_spec_optmax(int p, int kind) {
  (void)kind;   // (the compute role is the same for every proposal kernel: OPTMAX P doubles of x per lane)
  return (p >= 1 && p <= 3) ? 20 : (p <= 5 ? 10 : (p <= 7 ? 8 : 0));
}
FMH_HIDDEN const void* k_spec_normal(int p, int kind) {
#define SPEC_N(PV, OV) ((kind == 1) ? (const void*)mh_sweep_spec<PV, OV, 1> : (const void*)mh_sweep_spec<PV, OV, 2>)
  switch (p) {
    case 1: return SPEC_N(1, 20);
    case 2: return SPEC_N(2, 20);
    case 3: return SPEC_N(3, 20);
    case 4: return SPEC_N(4, 10);
    case 5: return SPEC_N(5, 10);
    case 6: return SPEC_N(6, 8);
    case 7: return SPEC_N(7, 8);
    default: return nullptr;
  }
#undef SPEC_N
}
}  // namespace fmh
