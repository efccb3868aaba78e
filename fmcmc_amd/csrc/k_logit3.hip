// k_logit3.hip -- mh_sweep_logit2<KIND> (mh_logit2.hpp): the observation-sharded logistic sweep of the normal / uniform proposal kernels,
// owners in the shadow of the grid-wide hand-overs (config C5)
#include "mh_tu.hpp"
#include "mh_spec.hpp"
#include "mh_logit2.hpp"

namespace fmh {
FMH_HIDDEN const void* k_logit2(int kind) {
  return kind == FMCMC_KERNEL_NORMAL ? (const void*)mh_sweep_logit2<FMCMC_KERNEL_NORMAL>
       : kind == FMCMC_KERNEL_NORMAL_REFLECTIVE ? (const void*)mh_sweep_logit2<FMCMC_KERNEL_NORMAL_REFLECTIVE> : nullptr;
}
FMH_HIDDEN size_t k_logit2_lds(int k) { return logit2_lds_bytes(k); }
FMH_HIDDEN const void* k_logit2a(int kind) {
  return kind == FMCMC_KERNEL_ADAPT ? (const void*)mh_sweep_logit2a<FMCMC_KERNEL_ADAPT>
       : kind == FMCMC_KERNEL_RAM ? (const void*)mh_sweep_logit2a<FMCMC_KERNEL_RAM> : nullptr;
}
FMH_HIDDEN size_t k_logit2a_lds() { return logit2a_lds_bytes(); }
}  // namespace fmh
