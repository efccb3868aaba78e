// k_spec_a.hip -- mh_sweep_spec<P, OPTMAX, KIND> (mh_spec.hpp) for kernel_adapt / kernel_ram (KIND 3, 4): 8 compute + 4 owner
// wavefronts meeting on LDS sequence words (config C3), one to four chains per workgroup
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_normal(int p, int kind);
FMH_HIDDEN const void* k_spec_mirror(int p, int kind);
FMH_HIDDEN const void* k_spec_w1(int p);   // k_spec_w1..4.hip: p = 8 .. 14
FMH_HIDDEN const void* k_spec_w2(int p);
FMH_HIDDEN const void* k_spec_w3(int p);
FMH_HIDDEN const void* k_spec_w4(int p);
const void* k_spec(int p, int kind) {
  if (p >= 8 && kind == FMCMC_KERNEL_ADAPT) return p <= 11 ? k_spec_w1(p) : k_spec_w2(p);
  if (p >= 8 && kind == FMCMC_KERNEL_RAM) return p <= 11 ? k_spec_w3(p) : k_spec_w4(p);
  if (kind == FMCMC_KERNEL_NORMAL || kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) return k_spec_normal(p, kind);
  if (kind == FMCMC_KERNEL_NMIRROR || kind == FMCMC_KERNEL_UMIRROR) return k_spec_mirror(p, kind);
  if (kind != FMCMC_KERNEL_ADAPT && kind != FMCMC_KERNEL_RAM) return nullptr;
#define SPEC_AD(PV, OV) ((kind == 3) ? (const void*)mh_sweep_spec<PV, OV, 3> : (const void*)mh_sweep_spec<PV, OV, 4>)
  switch (p) {
    case 0: return SPEC_AD(0, 20);   // (no covariate: the iid Normal family)
    case 1: return SPEC_AD(1, 20);
    case 2: return SPEC_AD(2, 20);
    case 3: return SPEC_AD(3, 20);
    case 4: return SPEC_AD(4, 10);
    case 5: return SPEC_AD(5, 10);
    case 6: return SPEC_AD(6, 8);
    case 7: return SPEC_AD(7, 8);
    default: return nullptr;
  }
#undef SPEC_AD
}
}  // namespace fmh
