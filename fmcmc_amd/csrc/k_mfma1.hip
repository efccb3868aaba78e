// k_mfma1.hip -- mh_sweep_mfma<1, NG, NS> (mh_mfma.hpp): kernel_normal / kernel_unif, the whole data set in operand registers
#include "mh_tu.hpp"
#include "mh_mfma.hpp"

namespace fmh {
FMH_HIDDEN const void* k_mfma_kv1(int ng, int ns, int big) {
#define MF_CASE(GV, SV) case SV: return big ? (const void*)mh_sweep_mfma<1, GV, SV, false, true> : (const void*)mh_sweep_mfma<1, GV, SV, false, false>;
#define MF_CASES10(GV) MF_CASE(GV, 1) MF_CASE(GV, 2) MF_CASE(GV, 3) MF_CASE(GV, 4) MF_CASE(GV, 5) MF_CASE(GV, 6) MF_CASE(GV, 7) MF_CASE(GV, 8) MF_CASE(GV, 9) MF_CASE(GV, 10)
#define MF_CASES20(GV) MF_CASES10(GV) MF_CASE(GV, 11) MF_CASE(GV, 12) MF_CASE(GV, 13) MF_CASE(GV, 14) MF_CASE(GV, 15) MF_CASE(GV, 16) MF_CASE(GV, 17) MF_CASE(GV, 18) MF_CASE(GV, 19) MF_CASE(GV, 20)
  if (ng == 1) { switch (ns) { MF_CASES20(1) default: return nullptr; } }
  if (ng == 2) { switch (ns) { MF_CASES10(2) default: return nullptr; } }
  return nullptr;
#undef MF_CASES20
#undef MF_CASES10
#undef MF_CASE
}
FMH_HIDDEN const void* k_mfma_kv2(int ng, int ns, int big);
const void* k_mfma(int kv, int ng, int ns, int big) { return kv == 1 ? k_mfma_kv1(ng, ns, big) : kv == 2 ? k_mfma_kv2(ng, ns, big) : nullptr; }
}  // namespace fmh
