// k_logit1.hip -- logistic-only instantiations of mh_sweep_kernel (mh_streamed.hpp), g table in LDS: the observation-sharded form (logit_shard) and the long-data form,
// variates drawn in the kernel (calls whose stream is not materialised: more than 1 GiB of it, single-parameter schemes)
#include "mh_tu.hpp"
#include "mh_streamed.hpp"

namespace fmh {
FMH_HIDDEN const void* k_logit_1(int cw, int kind) {
#define LK(CWV) ((kind == 1) ? (const void*)mh_sweep_kernel<CWV, -1, 2, 1, FMCMC_FAM_LOGISTIC, 1> : (kind == 2) ? (const void*)mh_sweep_kernel<CWV, -1, 2, 2, FMCMC_FAM_LOGISTIC, 1> \
               : (kind == 3) ? (const void*)mh_sweep_kernel<CWV, -1, 2, 3, FMCMC_FAM_LOGISTIC, 1> : (kind == 4) ? (const void*)mh_sweep_kernel<CWV, -1, 2, 4, FMCMC_FAM_LOGISTIC, 1> : nullptr)
  return cw == 1 ? LK(1) : cw == 2 ? LK(2) : cw == 4 ? LK(4) : nullptr;
#undef LK
}
}  // namespace fmh
