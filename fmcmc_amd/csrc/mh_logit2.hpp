// mh_logit2.hpp -- mh_sweep_logit2<KIND>: the observation-sharded logistic sweep of the normal / uniform proposal kernels (joint
// scheme, no fixed parameter, variates from a stream) with the owners' work IN THE SHADOW of the grid-wide hand-overs (round 5).
#pragma once

namespace {

// ==============================================================================================
// Why.  Config C5 on the general kernel (mh_sweep_kernel<4,-1,2,KIND,LOGISTIC>, tools/stamp_c5.py): of a 61 us step the slices'
// observations are 53, the two grid-wide hand-overs 3.2 + 4.3 -- and the owners' phases, each between workgroup barriers with seven
// waves waiting, another ~3.5: the variates' tile, the proposal, publish, the linear part of the closed form (a k-term fma chain),
// accept, rows.  None of that depends on the evaluation except the last subtraction and the compare.
// What.  The same decomposition (workgroup b of 256 = canonical lanes 2b, 2b + 1 of ALL chains, logit_shard; owner wave w of a
// workgroup = its chain w, lane = parameter; exchange tables sh_th / sh_part, the two-level arrival counters of shard_barrier),
// but a step is
//     publish theta1 | arrive 1 | [owners: sum_j b_j hs_j and the prior term of theta1, BOTH candidates of the next proposal
//     (theta1 + dz, theta0 + dz, reflected), the next log-uniform, the kept row of the step before] | wait 1 | logit_shard |
//     arrive 2 | wait 2 | gather, tree | f1 = lin - tot, compare, select | publish ...
// with the arrival and the poll on a wave that owns nothing (wave 7), so the owners' ~0.5 us of dependent work runs while the
// counters travel.  Same canonical lanes, same tree, same closed form (finish_logpost<LOGISTIC>), same rows: the bits are the
// general kernel's and the oracle's (tests/test_gpu_parity.py: every logistic test of the normal kernels with >= 1024 ... runs here).
// ==============================================================================================
constexpr int LG2_CW = 4;   // chains per workgroup (owner waves 0 .. 3)

// the two halves of shard_barrier (mh_common.hpp), the counters on the LAST wave's lane 0
__device__ __forceinline__ void lg2_arrive(unsigned* bar, unsigned epoch, bool fault) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);           // this thread's sc1 stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == NT - 64) {
    const unsigned ngroups = 8, gsize = gridDim.x / 8, g = blockIdx.x % ngroups;
    unsigned* top = bar + 8 * 32;
    unsigned* rel = bar + 9 * 32;
    if (!(fault && blockIdx.x == 1 && epoch == 3)) {
      const unsigned old = __hip_atomic_fetch_add(&bar[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == epoch * gsize) {
        const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == epoch * ngroups)
          for (unsigned q = 0; q < ngroups; q++) __hip_atomic_store(&rel[q * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__device__ __forceinline__ bool lg2_wait(unsigned* bar, unsigned epoch, bool fault) {
  bool ok = true;
  if (threadIdx.x == NT - 64) {
    const unsigned g = blockIdx.x % 8;
    unsigned* rel = bar + 9 * 32;
    unsigned spins = 0;
    const unsigned limit = fault ? 400000u : 20000000u;
    while (__hip_atomic_load(&rel[g * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > limit) { ok = false; break; }
    }
  }
  ok = !__syncthreads_or(ok ? 0 : 1);      // the verdict of the polling lane, for every thread of the workgroup
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  asm volatile("" ::: "memory");
  return ok;
}

template <int KIND>
__global__ __launch_bounds__(NT, 1) void mh_sweep_logit2(const SweepArgs A) {
  static_assert(KIND == FMCMC_KERNEL_NORMAL || KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE, "normal / uniform proposal kernels");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, nb = A.intercept + A.p;           // (k == nb for this family)
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  // ---- LDS: kernel constants, data-only sums, wave partials, then the g table
  double* s_mu = smem;
  double* s_scale = s_mu + k;
  double* s_lb = s_scale + k;
  double* s_ub = s_lb + k;
  double* s_hs = s_ub + k;
  double* s_part = s_hs + k;                           // [NW][LG2_CW]
  double* s_tab = logit_table_align(s_part + NW * LG2_CW);
  if (tid < k) {
    s_mu[tid] = A.mu[tid];
    s_scale[tid] = A.scale[tid];
    s_lb[tid] = A.lb[tid];
    s_ub[tid] = A.ub[tid];
    s_hs[tid] = (A.lg_hs && tid < nb) ? A.lg_hs[tid] : 0.0;
  }
  logit_stage_table(s_tab);
  logit_reset_turns(s_tab);
  __syncthreads();

  const long long cg0 = (long long)blockIdx.x * LG2_CW;
  const int ncw = (int)((A.nchains - cg0 < LG2_CW) ? (A.nchains - cg0 < 0 ? 0 : A.nchains - cg0) : LG2_CW);
  const bool owner = wave < ncw;
  const long long cl = cg0 + (owner ? wave : 0);
  const bool pl = lane < k;                            // a parameter lane
  const int jl = pl ? lane : 0;
  const int NCP = (int)A.nchains + SH_PAD;
  const bool fault = (A.debug & 512) != 0;
  constexpr unsigned LOST = 0x80000000u;
  unsigned epoch = 0;

  // ---- owner state: lane j <-> parameter j
  double th0 = (owner && pl) ? A.theta0[cl * k + jl] : 0.0, th1 = th0;
  double f0 = 0.0;
  int status = FMCMC_CHAIN_OK, thin_ctr = 0, nacc = 0;
  unsigned int bitword = 0, srow = 0;
  const double mu_l = s_mu[jl], sc_l = s_scale[jl], lb_l = s_lb[jl], ub_l = s_ub[jl];
  double* const out_s = A.samples + (cl * k + jl) * A.ldS;
  double* const out_d = A.draws ? A.draws + (cl * k + jl) * A.ldS : nullptr;
  double* const out_l = A.logpost ? A.logpost + cl * A.ldS : nullptr;
  const double* const z_lane = A.fed_z + (cl * nsteps) * A.kz + jl;       // (kz == k: no fixed parameter)
  const double* const lu_row = A.fed_logu + cl * nsteps;
  // the row decided last, stored in the shadow of the next hand-over
  bool st_row = false;
  double st_th0 = 0.0, st_dr = 0.0, st_f = 0.0;
  // what the decision of the evaluation under way needs and the evaluation does not enter
  double lin1 = 0.0, pri1 = 0.0, cand_a = 0.0, cand_r = 0.0, lu = 0.0;

  auto publish = [&]() {
    if (owner && pl) sh_store(&A.sh_th[(long long)lane * NCP + cl], th1);
  };
  publish();

  for (int v = 1; v <= nsteps; v++) {                  // v = 1: the initial state; v >= 2: the proposal of loop step v
    // ================= hand-over 1: the proposals of all chains =================
    const bool sync_on = !(A.debug & 32) && !(epoch & LOST);
    if (sync_on) lg2_arrive(A.sh_bar, ++epoch, fault); else __syncthreads();
    // ---- in its shadow, the owners
    if (owner) {
      // variates of loop step v + 1 (row v of the stream) and the log-uniform of decision v (row v - 1); clamped rows are not used
      const double zn = pl ? z_lane[(long long)(v < nsteps ? v : nsteps - 1) * A.kz] : 0.0;
      const double lun = lu_row[v >= 2 ? v - 1 : 0];
      if (st_row) {                                    // the row decided at v - 1
        if (pl) {
          out_s[srow] = st_th0;
          if (out_d) out_d[srow] = st_dr;
        }
        if (out_l && lane == 0) out_l[srow] = st_f;
        srow += 1;
        st_row = false;
      }
      // closed form without its total: sum_j b_j hs_j, the prior term (finish_logpost<LOGISTIC>: the same chains, the same bits)
      double lin = 0.0, ss = 0.0;
      for (int j = 0; j < nb; j++) {
        const double bj = readlane_d(th1, j);
        lin = fmh_fma(bj, s_hs[j], lin);
        ss = fmh_fma(bj, bj, ss);
      }
      lin1 = lin;
      pri1 = (A.prior_div != 0.0) ? ss / A.prior_div : 0.0;
      // both candidates of the next proposal
      const double dz = mu_l + sc_l * zn;
      double ca = th1 + dz, cr = th0 + dz;
      if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) { ca = reflect1(ca, lb_l, ub_l); cr = reflect1(cr, lb_l, ub_l); }
      cand_a = ca; cand_r = cr; lu = lun;
    }
    if (sync_on) { if (!lg2_wait(A.sh_bar, epoch, fault)) epoch |= LOST; }
    // ================= the slices' observations for all chains =================
    eval_sharded_logit_step<2>(A, s_tab);
    // ================= hand-over 2: the lane partials =================
    const bool sync2 = !(A.debug & 32) && !(epoch & LOST);
    if (sync2) {
      lg2_arrive(A.sh_bar, ++epoch, fault);
      if (!lg2_wait(A.sh_bar, epoch, fault)) epoch |= LOST;
    } else {
      __syncthreads();
    }
    // ---- thread = canonical lane: its partial of this workgroup's chains, levels 1 .. 32 of the tree
    {
      double acc[LG2_CW];
#pragma unroll
      for (int c = 0; c < LG2_CW; c++)
        acc[c] = (c < ncw && !(A.debug & 128)) ? sh_load(A.sh_part + ((unsigned int)(cg0 + c) * (unsigned int)(NT + SH_PAD) + (unsigned int)tid)) : 1.0;
#pragma unroll
      for (int c = 0; c < LG2_CW; c++) {
        const double w = wave_xor_sum(acc[c]);
        if (lane == 0) s_part[wave * LG2_CW + c] = w;
      }
    }
    lds_barrier();
    // ================= owners: the decision (R/mcmc.R:754-778) =================
    if (owner) {
      const int c = wave;
      const double w0 = s_part[0 * LG2_CW + c], w1 = s_part[1 * LG2_CW + c], w2 = s_part[2 * LG2_CW + c], w3 = s_part[3 * LG2_CW + c];
      const double w4 = s_part[4 * LG2_CW + c], w5 = s_part[5 * LG2_CW + c], w6 = s_part[6 * LG2_CW + c], w7 = s_part[7 * LG2_CW + c];
      const double tot = ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7));  // levels 64, 128, 256
      double f1 = lin1 - tot;
      if (A.prior_div != 0.0) f1 = f1 - pri1;
      if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
      if (v == 1) {
        f0 = f1;
        if (1 > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; st_dr = th1; st_f = f1; } }
        th1 = cand_r;                                  // (both candidates are theta0 + dz before the first decision)
      } else if (status == FMCMC_CHAIN_OK) {
        const int i = v;
        if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
        const double ratio = f1 - f0;
        if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
        if (status != FMCMC_CHAIN_OK) {
          if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
          if (pl) A.status_theta[cl * k + lane] = th1;
        } else {
          const bool acc = lu < ratio;
          const double dr = th1;
          if (acc) {
            th0 = th1;
            f0 = f1;
            nacc += 1;
            bitword |= (1u << ((i - 1) & 31));
          }
          if (i > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; st_dr = dr; st_f = f1; } }
          th1 = acc ? cand_a : cand_r;
        }
      }
      if (v >= 2 && A.accept_bits && lane == 0 && (((v - 1) & 31) == 31 || v == nsteps)) {
        A.accept_bits[cl * (long long)((nsteps + 31) >> 5) + ((v - 1) >> 5)] = bitword;
        bitword = 0;
      }
      if (v < nsteps) publish();
    }
  }
  // ---- the last row, state
  if (owner) {
    if (st_row) {
      if (pl) {
        out_s[srow] = st_th0;
        if (out_d) out_d[srow] = st_dr;
      }
      if (out_l && lane == 0) out_l[srow] = st_f;
    }
    if (pl) A.theta0[cl * k + lane] = th0;
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      if (epoch & LOST) { A.status[cl] = FMCMC_CHAIN_SYNC_TIMEOUT; A.status_step[cl] = 0; }
      else if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    }
  }
}

// ---- kernel_adapt / kernel_ram on the same sweep: the register owner of mh_spec.hpp (spec_owner_adaptive_reg<.., LOGISTIC>) meets the
// evaluation through Logit2Sync -- total() IS the collective part of a step (hand-over 1 with the owner's preparation in its shadow,
// logit_shard, hand-over 2, gather and tree), the waves that own nothing run the same calls in a loop of their own.
template <class COLL>
struct Logit2Sync {
  static constexpr bool PREP_EARLY = true;
  COLL& coll; const double* s_part; const double* s_th1; const SweepArgs& A; int myc; long long cl; int ncp;
  template <class F> __device__ __forceinline__ double total(int, F&& prep) const {
    coll(prep);
    const int c = myc;
    const double w0 = s_part[0 * LG2_CW + c], w1 = s_part[1 * LG2_CW + c], w2 = s_part[2 * LG2_CW + c], w3 = s_part[3 * LG2_CW + c];
    const double w4 = s_part[4 * LG2_CW + c], w5 = s_part[5 * LG2_CW + c], w6 = s_part[6 * LG2_CW + c], w7 = s_part[7 * LG2_CW + c];
    return ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7));  // levels 64, 128, 256
  }
  __device__ __forceinline__ void publish(int) const {       // (the owner has just written its proposal to s_th1: this lane's own element)
    const int lane = threadIdx.x & 63;
    if (lane < A.k) sh_store(&A.sh_th[(long long)lane * ncp + cl], s_th1[myc * PIPE_KMAX + lane]);
  }
  __device__ __forceinline__ void final() const {}
};

template <int KIND>
__global__ __launch_bounds__(NT, 1) void mh_sweep_logit2a(const SweepArgs A) {
  static_assert(KIND == FMCMC_KERNEL_ADAPT || KIND == FMCMC_KERNEL_RAM, "the adaptive proposal kernels");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k;
  const int nsteps = (int)A.nsteps;
  double* s_th1 = smem;                                // [LG2_CW][PIPE_KMAX] the owners' proposals (spec_owner_adaptive_reg writes them here)
  double* s_part = s_th1 + LG2_CW * PIPE_KMAX;         // [NW][LG2_CW]
  double* s_tab = logit_table_align(s_part + NW * LG2_CW);
  logit_stage_table(s_tab);
  logit_reset_turns(s_tab);
  const long long cg0 = (long long)blockIdx.x * LG2_CW;
  const int ncw = (int)((A.nchains - cg0 < LG2_CW) ? (A.nchains - cg0 < 0 ? 0 : A.nchains - cg0) : LG2_CW);
  const bool owner = wave < ncw;
  const long long cl = cg0 + (owner ? wave : 0);
  const int NCP = (int)A.nchains + SH_PAD;
  const bool fault = (A.debug & 512) != 0;
  constexpr unsigned LOST = 0x80000000u;
  unsigned epoch = 0;
  if (owner && lane < k) {                             // the initial state is the first "proposal"
    const double t = A.theta0[cl * k + lane];
    s_th1[wave * PIPE_KMAX + lane] = t;
    sh_store(&A.sh_th[(long long)lane * NCP + cl], t);
  }
  __syncthreads();
  // the collective part of a step; `prep` runs between the arrival and the wait of hand-over 1 (owners: what the decision needs and
  // the evaluation does not enter)
  auto coll = [&](auto&& prep) {
    const bool sync_on = !(A.debug & 32) && !(epoch & LOST);
    if (sync_on) lg2_arrive(A.sh_bar, ++epoch, fault); else __syncthreads();
    prep();
    if (sync_on) { if (!lg2_wait(A.sh_bar, epoch, fault)) epoch |= LOST; }
    eval_sharded_logit_step<2>(A, s_tab);
    const bool sync2 = !(A.debug & 32) && !(epoch & LOST);
    if (sync2) {
      lg2_arrive(A.sh_bar, ++epoch, fault);
      if (!lg2_wait(A.sh_bar, epoch, fault)) epoch |= LOST;
    } else {
      __syncthreads();
    }
    double acc[LG2_CW];
#pragma unroll
    for (int c = 0; c < LG2_CW; c++)
      acc[c] = (c < ncw && !(A.debug & 128)) ? sh_load(A.sh_part + ((unsigned int)(cg0 + c) * (unsigned int)(NT + SH_PAD) + (unsigned int)tid)) : 1.0;
#pragma unroll
    for (int c = 0; c < LG2_CW; c++) {
      const double w = wave_xor_sum(acc[c]);
      if (lane == 0) s_part[wave * LG2_CW + c] = w;
    }
    lds_barrier();
  };
  if (owner) {
    Logit2Sync<decltype(coll)> sync{coll, s_part, s_th1, A, wave, cl, NCP};
    spec_owner_adaptive_reg<KIND, 0, decltype(sync), false, FMCMC_FAM_LOGISTIC>(A, wave, (int)cl, s_th1, sync);
    if ((epoch & LOST) && lane == 0) { A.status[cl] = FMCMC_CHAIN_SYNC_TIMEOUT; A.status_step[cl] = 0; }
  } else {
    for (int v = 1; v <= nsteps; v++) coll([]() {});
  }
}

size_t logit2a_lds_bytes() { return sizeof(double) * (size_t)(LG2_CW * PIPE_KMAX + NW * LG2_CW + LG_LDS_TAIL + LG_LDS_DOUBLES); }

size_t logit2_lds_bytes(int k) { return sizeof(double) * (size_t)(5 * k + NW * LG2_CW + LG_LDS_TAIL + LG_LDS_DOUBLES); }

}  // namespace
