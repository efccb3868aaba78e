// k_logit0.hip -- logistic-only instantiations of mh_sweep_kernel (mh_streamed.hpp), g table in LDS: the chain-sharded loop (logit_partials)
#include "mh_tu.hpp"
#include "mh_streamed.hpp"

namespace fmh {
FMH_HIDDEN const void* k_logit_0(int cw, int kind) {
#define LK(CWV) ((kind == 1) ? (const void*)mh_sweep_kernel<CWV, -1, 0, 1, FMCMC_FAM_LOGISTIC, 1> : (kind == 2) ? (const void*)mh_sweep_kernel<CWV, -1, 0, 2, FMCMC_FAM_LOGISTIC, 1> \
               : (kind == 3) ? (const void*)mh_sweep_kernel<CWV, -1, 0, 3, FMCMC_FAM_LOGISTIC, 1> : (kind == 4) ? (const void*)mh_sweep_kernel<CWV, -1, 0, 4, FMCMC_FAM_LOGISTIC, 1> : nullptr)
  return cw == 1 ? LK(1) : cw == 2 ? LK(2) : cw == 4 ? LK(4) : nullptr;
#undef LK
}
FMH_HIDDEN const void* k_logit_1(int cw, int kind);
FMH_HIDDEN const void* k_logit_2(int cw, int kind);
const void* k_logit(int cw, int sharded, int kind) { return sharded == 2 ? k_logit_2(cw, kind) : sharded ? k_logit_1(cw, kind) : k_logit_0(cw, kind); }
}  // namespace fmh
