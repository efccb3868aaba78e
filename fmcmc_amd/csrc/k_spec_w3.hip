// k_spec_w3.hip -- mh_sweep_spec<P, 4, 4> (mh_spec.hpp): kernel_ram with 8 .. 11 covariates on up to 2048 observations (four slots of P doubles
// per compute lane), the register owner at the compile-time width k = P + 2 <= 16
#include "mh_tu.hpp"
#include "mh_spec.hpp"

namespace fmh {
FMH_HIDDEN const void* k_spec_w3(int p) {
  switch (p) {
    case 8: return (const void*)mh_sweep_spec<8, 4, 4>;
    case 9: return (const void*)mh_sweep_spec<9, 4, 4>;
    case 10: return (const void*)mh_sweep_spec<10, 4, 4>;
    case 11: return (const void*)mh_sweep_spec<11, 4, 4>;
    default: return nullptr;
  }
}
}  // namespace fmh
