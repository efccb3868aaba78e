// k_spec_l2.hip -- mh_sweep_spec<P, OPTMAX, KIND, LOGISTIC> (mh_spec.hpp): the wave-specialised sweep for the logistic family (round 5), kernel_adapt / kernel_ram
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_lw1(int p);   // k_spec_lw1/2.hip: p = 8 .. 15
FMH_HIDDEN const void* k_spec_lw2(int p);
FMH_HIDDEN const void* k_spec_logit_a(int p, int kind) {
  if (p >= 8) return kind == 3 ? k_spec_lw1(p) : k_spec_lw2(p);
#define SPEC_L(PV, OV) ((kind == 3) ? (const void*)mh_sweep_spec<PV, OV, 3, FMCMC_FAM_LOGISTIC> : (const void*)mh_sweep_spec<PV, OV, 4, FMCMC_FAM_LOGISTIC>)
  switch (p) {
    case 1: return SPEC_L(1, 20);
    case 2: return SPEC_L(2, 20);
    case 3: return SPEC_L(3, 20);
    case 4: return SPEC_L(4, 10);
    case 5: return SPEC_L(5, 10);
    case 6: return SPEC_L(6, 8);
    case 7: return SPEC_L(7, 8);
    default: return nullptr;
  }
#undef SPEC_L
}
}  // namespace fmh
