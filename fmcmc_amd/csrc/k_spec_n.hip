// k_spec_n.hip -- mh_sweep_spec<P, 20, KIND> (mh_spec.hpp) for the normal / uniform kernels (KIND 1, 2) at p = 1, 3: the VALU
// partner of mh_sweep_mfma (knob mfma=0), a second implementation of the same sweep for the parity tests
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
int k_spec_optmax(int p, int kind) {
  // (the compute role is the same for every proposal kernel: OPTMAX P doubles of x per lane; 8 .. 14 covariates: the adaptive kernels only)
  return (p >= 0 && p <= 3) ? 20 : (p <= 5 ? 10 : (p <= 7 ? 8 : ((p <= 15 && (kind == FMCMC_KERNEL_ADAPT || kind == FMCMC_KERNEL_RAM || kind == FMCMC_KERNEL_NMIRROR || kind == FMCMC_KERNEL_UMIRROR)) ? 4 : 0)));
}
FMH_HIDDEN const void* k_spec_normal(int p, int kind) {
#define SPEC_N(PV, OV) ((kind == 1) ? (const void*)mh_sweep_spec<PV, OV, 1> : (const void*)mh_sweep_spec<PV, OV, 2>)
  switch (p) {
    case 1: return SPEC_N(1, 20);
    case 3: return SPEC_N(3, 20);
    default: return nullptr;
  }
#undef SPEC_N
}
}  // namespace fmh
