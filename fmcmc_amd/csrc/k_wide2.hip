// k_wide2.hip -- mh_sweep_wide2<KIND, NMT> (mh_wide2.hpp: wide models, observation-sharded dataflow form) and mh_sweep_bigk
// (mh_bigk.hpp: more parameters than a wavefront has lanes)
#define FMH_WITH_BIGK_KERNEL
#include "mh_tu.hpp"
#include "mh_streamed.hpp"
#include "mh_wide2.hpp"
#include "mh_bigk.hpp"

namespace fmh {
const void* k_wide2(int kind, int nmt) {
#define W2K(KV) ((nmt == 1) ? (const void*)mh_sweep_wide2<KV, 1> : (nmt == 2) ? (const void*)mh_sweep_wide2<KV, 2> : (nmt == 3) ? (const void*)mh_sweep_wide2<KV, 3> : nullptr)
  return (kind == 1) ? W2K(1) : (kind == 2) ? W2K(2) : (kind == 4) ? W2K(4) : nullptr;
#undef W2K
}
const void* k_bigk() { return (const void*)mh_sweep_bigk; }
}  // namespace fmh
