// k_wide.hip -- wide Gaussian linear models on mh_sweep_kernel (mh_streamed.hpp): one family and one proposal kernel compiled
// in; OPT = 0 the chain-sharded loop, OPT = 2 | 4 the observation-sharded evaluation (canonical lanes per workgroup), and the
// long-data form (one chain per workgroup, OPT = 2, every proposal kernel)
#include "mh_tu.hpp"
#include "mh_streamed.hpp"

namespace fmh {
const void* k_wide(int cw, int lpw, int kind) {
#define WK(CWV, LV) ((kind == 1) ? (const void*)mh_sweep_kernel<CWV, -1, LV, 1, FMCMC_FAM_GAUSSIAN_LINREG>   \
                   : (kind == 2) ? (const void*)mh_sweep_kernel<CWV, -1, LV, 2, FMCMC_FAM_GAUSSIAN_LINREG>   \
                   : (kind == 4) ? (const void*)mh_sweep_kernel<CWV, -1, LV, 4, FMCMC_FAM_GAUSSIAN_LINREG> : nullptr)
  if (cw == 1 && lpw == 2 && kind == 3) return (const void*)mh_sweep_kernel<1, -1, 2, 3, FMCMC_FAM_GAUSSIAN_LINREG>;   // (long-data form)
  if (cw == 1) return lpw == 0 ? WK(1, 0) : lpw == 2 ? WK(1, 2) : lpw == 4 ? WK(1, 4) : nullptr;
  if (cw == 2) return lpw == 0 ? WK(2, 0) : lpw == 2 ? WK(2, 2) : lpw == 4 ? WK(2, 4) : nullptr;
#undef WK
  return nullptr;
}
}  // namespace fmh
