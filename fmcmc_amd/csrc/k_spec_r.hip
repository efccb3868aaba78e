// k_spec_r.hip -- mh_sweep_spec<P, OPTMAX, 3, FAM, RING = true> (mh_spec.hpp): kernel_adapt(freq = 2 .. 8, bw = 0) on the register owner
// with the LDS ring of the chain's last rows, linear and logistic model
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_ring(int p, int logistic) {
#define SPEC_R(PV, OV) (logistic ? (const void*)mh_sweep_spec<PV, OV, FMCMC_KERNEL_ADAPT, FMCMC_FAM_LOGISTIC, true> \
                                 : (const void*)mh_sweep_spec<PV, OV, FMCMC_KERNEL_ADAPT, FMCMC_FAM_GAUSSIAN_LINREG, true>)
  switch (p) {
    case 0: return logistic ? nullptr : (const void*)mh_sweep_spec<0, 20, FMCMC_KERNEL_ADAPT, FMCMC_FAM_GAUSSIAN_LINREG, true>;
    case 1: return SPEC_R(1, 20);
    case 2: return SPEC_R(2, 20);
    case 3: return SPEC_R(3, 20);
    case 4: return SPEC_R(4, 10);
    case 5: return SPEC_R(5, 10);
    case 6: return SPEC_R(6, 8);
    case 7: return SPEC_R(7, 8);
    default: return nullptr;
  }
#undef SPEC_R
}
}  // namespace fmh
