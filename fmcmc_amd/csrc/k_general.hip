// k_general.hip -- the all-family kernel mh_sweep_kernel<CW, -1, 0, 0> and the two register-resident shapes (mh_streamed.hpp)
#include "mh_tu.hpp"
#include "mh_streamed.hpp"

namespace fmh {
const void* k_general(int cw) {
  switch (cw) {
    case 1: return (const void*)mh_sweep_kernel<1, -1, 0, 0>;
    case 2: return (const void*)mh_sweep_kernel<2, -1, 0, 0>;
    case 4: return (const void*)mh_sweep_kernel<4, -1, 0, 0>;
    case 8: return (const void*)mh_sweep_kernel<8, -1, 0, 0>;
    default: return nullptr;
  }
}
const void* k_resident(int p, int kind) {
#define RES_K(PV, OV) ((kind == 1) ? (const void*)mh_sweep_kernel<4, PV, OV, 1> : (kind == 2) ? (const void*)mh_sweep_kernel<4, PV, OV, 2> \
                     : (kind == 3) ? (const void*)mh_sweep_kernel<4, PV, OV, 3> : (const void*)mh_sweep_kernel<4, PV, OV, 4>)
  if (p == 1) return RES_K(1, 4);
  if (p == 3) return RES_K(3, 20);
#undef RES_K
  return nullptr;
}
}  // namespace fmh
