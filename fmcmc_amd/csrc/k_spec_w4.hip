// k_spec_w4.hip -- mh_sweep_spec<P, 4, 4> (mh_spec.hpp): kernel_ram with 12 .. 14 covariates on up to 2048 observations (four slots of P doubles
// per compute lane), the register owner at the compile-time width k = P + 2 <= 16
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_w4(int p) {
  switch (p) {
    case 12: return (const void*)mh_sweep_spec<12, 4, 4>;
    case 13: return (const void*)mh_sweep_spec<13, 4, 4>;
    case 14: return (const void*)mh_sweep_spec<14, 4, 4>;
    default: return nullptr;
  }
}
}  // namespace fmh
