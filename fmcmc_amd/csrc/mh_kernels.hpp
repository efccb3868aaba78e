// mh_kernels.hpp -- kernel look-ups: shape -> the host handle of the instantiation that runs it (nullptr: none compiled).
// launch_sweep (mh_engine.hip) launches the handle with hipLaunchKernel / hipLaunchCooperativeKernel; every kernel takes ONE
// argument, the SweepArgs of the launch, by value.
#pragma once

#define FMH_HIDDEN __attribute__((visibility("hidden")))

namespace fmh {
// k_general.hip: mh_sweep_kernel<CW, -1, 0, 0> (every family / proposal kernel / scheme), cw = 1, 2, 4, 8;
//                the register-resident shapes mh_sweep_kernel<4, P, OPT, KIND>: (p, opt) = (1, 4), (3, 20), kind 1..4
FMH_HIDDEN const void* k_general(int cw);
FMH_HIDDEN const void* k_resident(int p, int kind);
// k_wide.hip: wide linear models mh_sweep_kernel<CW, -1, LPW, KIND, LINREG>: cw = 1, 2; lpw = 0 (chain-sharded), 2, 4 (observation-
//             sharded, cooperative); kind = 1, 2, 4 -- and the long-data form <1, -1, 2, KIND, LINREG>, kind 1..4
FMH_HIDDEN const void* k_wide(int cw, int lpw, int kind);
// k_logit*.hip: logistic-only instantiations mh_sweep_kernel<CW, -1, OPT, KIND, LOGISTIC, 1>: cw = 1, 2, 4; sharded = 0 | 1 | 2 (OPT = 0 | 2 | 2; 2: variates from a materialised stream only)
FMH_HIDDEN const void* k_logit(int cw, int sharded, int kind);
// k_logit3.hip: mh_sweep_logit2<KIND> (kind 1, 2): the observation-sharded sweep with the owners in the shadow of the hand-overs
FMH_HIDDEN const void* k_logit2(int kind);
FMH_HIDDEN size_t k_logit2_lds(int k);
FMH_HIDDEN const void* k_logit2a(int kind);     // mh_sweep_logit2a<KIND> (kind 3, 4; k <= 8, no fixed parameter, unbounded kernel_ram)
FMH_HIDDEN size_t k_logit2a_lds();
// k_mfma*.hip: mh_sweep_mfma<KV, NG, NS, false, BIG>, and the streamed-operand form <KV, NG, NSRES, false, BIG, true>
FMH_HIDDEN const void* k_mfma(int kv, int ng, int ns, int big);
FMH_HIDDEN const void* k_mfma_ext(int kv, int ng, int nsres, int big);
// k_mfma_ad.hip: mh_sweep_mfma_ad<KIND, NG, KX, BND, NSV>: kx = compile-time row count (5, 9), 0 (k <= 8), -1 (matrices in
//                LDS), -2 (mirror kernels); shrt = 1: one resident slot
FMH_HIDDEN const void* k_mfma_ad(int kind, int ng, int kx, int bnd, int shrt);
// k_spec.hip: mh_sweep_spec<P, OPTMAX, KIND>
FMH_HIDDEN const void* k_spec(int p, int kind);
FMH_HIDDEN const void* k_spec_ring(int p, int logistic);   // k_spec_r.hip: kernel_adapt(freq = 2 .. 8)
FMH_HIDDEN int k_spec_optmax(int p, int kind);
// k_spec_l*.hip: mh_sweep_spec<P, OPTMAX, KIND, LOGISTIC>: p = 1 .. 7, kind 1 .. 4
FMH_HIDDEN const void* k_spec_logit(int p, int kind);
FMH_HIDDEN size_t k_spec_logit_lds(int adaptive);
// k_lat*.hip: mh_sweep_lat<KIND, P, OPTMAX> (kind 1, 2; p = 1 .. 7; the slot counts of k_spec_optmax)
FMH_HIDDEN const void* k_lat(int p, int kind);
// k_lat_l*.hip: mh_sweep_lat<KIND, P, OPTMAX, LOGISTIC> (kind 1, 2; p = 1 .. 7)
FMH_HIDDEN const void* k_lat_logit(int p, int kind);
FMH_HIDDEN size_t k_lat_logit_lds();
// k_wide2.hip: mh_sweep_wide2<KIND, NMT> (kind 1, 2, 4; nmt 1..3) and mh_sweep_bigk
FMH_HIDDEN const void* k_wide2(int kind, int nmt);
FMH_HIDDEN const void* k_bigk();
}  // namespace fmh
