// k_mfma_ext.hip -- mh_sweep_mfma<KV, NG, NSRES, false, BIG, EXT = true> (mh_mfma.hpp): NSRES observation slots in operand
// registers, the rest streamed from an operand-order copy every step (any n; 8 .. 15 covariates as three / four operand groups)
#include "mh_tu.hpp"
#include "mh_mfma.hpp"

namespace fmh {
const void* k_mfma_ext(int kv, int ng, int nsres, int big) {
#define MF_EXT(GV, SV) if (ng == GV && nsres == SV) return kv == 1 ? (big ? (const void*)mh_sweep_mfma<1, GV, SV, false, true, true> : (const void*)mh_sweep_mfma<1, GV, SV, false, false, true>) \
                                                          : kv == 2 ? (big ? (const void*)mh_sweep_mfma<2, GV, SV, false, true, true> : (const void*)mh_sweep_mfma<2, GV, SV, false, false, true>) : nullptr;
  MF_EXT(1, 16) MF_EXT(2, 8) MF_EXT(3, 4) MF_EXT(3, 1) MF_EXT(4, 2) MF_EXT(4, 1)
#undef MF_EXT
  return nullptr;
}
}  // namespace fmh
