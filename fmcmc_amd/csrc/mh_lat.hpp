// mh_lat.hpp -- mh_sweep_lat<KIND, P>: the LATENCY form of the sweep for fewer than four chains per compute unit (round 5).
#pragma once

namespace {

// ==============================================================================================
// Why.  The reference scales a FIXED number of chains over its workers (R/mcmc.R:536-641: nchains over min(nchains, cores)
// PSOCK processes), and a sharded call leaves a GPU nchains / G of them.  The kernels of full launches put four chains on a
// workgroup (one workgroup per CU): a step of C2's shape costs the same 2 us at 64 chains as at 1024, and the owners /
// evaluators there meet twice per step (two barriers, or two LDS flag hand-overs), which is most of a step once the
// evaluation is a single chain's.
//
// What.  One to three chains per workgroup, 8 wavefronts = the 512 canonical lanes, x AND y of a lane's observations in
// VGPRs, and the chain state REPLICATED in every wavefront: lane 16 c + j of every wave holds parameter j of the workgroup's
// chain c (theta0 and the proposal theta1; f0 and the status are uniform over a 16-lane row).  A step is
//     evaluate theta1 of every chain (VALU, the canonical lanes' fma chains) | in-wave tree levels 1..32 | ONE barrier |
//     every wave: levels 64..256 over 8 values, closed form, accept, select the next proposal -- the same instructions on the
//     same inputs in all eight waves, so nothing has to be published and no second hand-over exists;
// what the decision needs and the evaluation does not enter is prepared by DUTY waves on four different SIMDs while they
// evaluate, and handed over through LDS at the same barrier (double-buffered by the step's parity):
//     wave 0  the keeper: rows of ans / draws / logpost, accept bitmap and counts, chain status, the state at the end;
//     wave 1  n (log sigma + ln sqrt 2 pi) of the proposal under evaluation (the logarithm: ~60 dependent instructions);
//     wave 2  BOTH candidates of the next proposal -- theta1 + dz (should this one be accepted) and theta0 + dz -- reflected
//             where the kernel reflects, and the log-uniform of the pending decision: the only wave that reads the stream;
//     wave 3  sigma^2 and the reciprocal half of the division (div_recip / div_finish: bit for bit the quotient).
// The ROLE of a wave is a template parameter of the step loop (one copy of the loop per role): a wave carries the registers
// of its own duty only -- 160 of its 256 VGPRs are data at C2's shape -- and the logarithm of wave 1 sits in the same basic
// block as its evaluation, so that the scheduler interleaves the two dependency chains.
// The decision itself is then ~30 instructions: fold, three fmas, a compare, selects.
// Same canonical lanes, same tree, same closed form as every other kernel: the bits do not depend on the form
// (tests/test_gpu_parity.py, test_latency_form_*).
// ==============================================================================================
constexpr int LAT_ROWS = 4;                      // 16-lane rows of a wavefront = chains a workgroup can hold (the dispatcher uses 1..3)
constexpr int LAT_FOLD = 0;                      // [2][LAT_ROWS][NW]   per-wave sums of every chain (levels 1..32 done)
constexpr int LAT_PREP = LAT_FOLD + 2 * LAT_ROWS * NW;   // [2][4][LAT_ROWS]  nt1, flag of nt1 (1: sigma regular), sigma^2, 1 / sigma^2 (0: no fast division)
constexpr int LAT_LU = LAT_PREP + 2 * 4 * LAT_ROWS;      // [2][LAT_ROWS]     log-uniform of the pending decision
constexpr int LAT_CAND = LAT_LU + 2 * LAT_ROWS;          // [2][2][64]        next proposal if accepted / if rejected, lane-wise
constexpr int LAT_PAR = LAT_CAND + 2 * 2 * 64;           // [4][16]           mu, scale, lb, ub
constexpr int LAT_SINK = LAT_PAR + 64;                   // [64]              where the lanes without a result write
constexpr int LAT_PLAN = LAT_SINK + 64;                  // [66] ints         single-parameter schemes: a lane's rank among the free parameters, their number, the first
constexpr int LAT_LDS_DOUBLES = LAT_PLAN + 34;

enum { LAT_GENERIC = 0, LAT_KEEPER = 1, LAT_LOG = 2, LAT_CANDS = 3, LAT_RECIP = 4, LAT_CANDS_S = 5 /* the candidate wave under a single-parameter scheme */ };

template <int KIND, int P, int OPT, int ROLE, int FAM = FMCMC_FAM_GAUSSIAN_LINREG>
__device__ __forceinline__ void lat_steps(const SweepArgs& A, double* smem) {
  // FAM = LOGISTIC (round 5): the same form for the logistic family -- per observation 64 eta from the scaled coefficients and
  // g(|eta|) off the table in LDS (behind this kernel's block), added in slot order; y is not read; wave 1's duty is sum_j b_j hs_j and
  // the prior term instead of the logarithm, wave 3 has none; the decision is lin - total - prior.
  constexpr bool LG = FAM == FMCMC_FAM_LOGISTIC;
  constexpr bool CANDS = ROLE == LAT_CANDS || ROLE == LAT_CANDS_S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz, ic = A.intercept;
  const int nsteps = (int)A.nsteps;
  const int cwl = A.spec_cw;
  const long long cg0 = (long long)blockIdx.x * cwl;
  const int ncw = (int)((A.nchains - cg0 < cwl) ? (A.nchains - cg0) : cwl);
  double* s_fold = smem + LAT_FOLD;
  double* s_prep = smem + LAT_PREP;
  double* s_lu = smem + LAT_LU;
  double* s_cand = smem + LAT_CAND;
  const double* s_par = smem + LAT_PAR;
  double* s_sink = smem + LAT_SINK;

  // ---- this lane's observations: canonical lane tid, slots s = 0 .. OPT - 1 (observation tid + 512 s)
  double xr[OPT][P > 0 ? P : 1], yr[OPT];
  double wlast = 1.0, wprev = 1.0;   // validity of the last two slots (OPT = the slot count rounded up to even)
#pragma unroll
  for (int s = 0; s < OPT; s++) {
    const long long i = (long long)tid + (long long)NT * s;
    const bool valid = i < A.n;
#pragma unroll
    for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
    yr[s] = (valid && !LG) ? A.y[i] : 0.0;
    if (s == OPT - 1) wlast = valid ? 1.0 : 0.0;
    if (s == OPT - 2) wprev = valid ? 1.0 : 0.0;
  }

  // ---- replicated chain state: lane 16 c + j <-> parameter j of chain c
  const int row = lane >> 4, jj = lane & 15;
  const bool rowact = row < ncw;
  const bool act = rowact && jj < k;                // a parameter of a chain of this workgroup
  const long long cl = cg0 + (rowact ? row : 0);    // (idle rows shadow chain 0 of the workgroup and never write)
  const int jl = (jj < k) ? jj : 0;
  double th0 = act ? A.theta0[cl * k + jj] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  int status = (A.win_cont && rowact) ? A.status[cl] : FMCMC_CHAIN_OK;
  const double dn = (double)A.n;

  // ---- keeper: rows, bitmap, counts
  int nacc = 0, thin_ctr = A.thin_ctr0;
  unsigned int bitword = 0, srow8 = 0;
  char* s_ptr = nullptr; char* d_ptr = nullptr; char* l_ptr = nullptr;
  if constexpr (ROLE == LAT_KEEPER) {
    s_ptr = reinterpret_cast<char*>(A.samples) + ((cl * k + jl) * A.ldS) * 8;
    d_ptr = A.draws ? reinterpret_cast<char*>(A.draws) + ((cl * k + jl) * A.ldS) * 8 : nullptr;
    l_ptr = A.logpost ? reinterpret_cast<char*>(A.logpost) + (cl * A.ldS) * 8 : nullptr;
  }
  auto flush_bits = [&](int i) {   // (bits_stride: words per chain of the whole call's bitmap, set by launch_sweep for every launch)
    if constexpr (ROLE == LAT_KEEPER) {
      if (A.accept_bits && rowact && jj == 0) {
        unsigned int* w = A.accept_bits + (cl * A.bits_stride + ((i - 1) >> 5));
        // the first word of a continuation window also holds the last bit of the window before it
        *w = (A.win_cont && i <= 32) ? (*w | bitword) : bitword;
      }
      bitword = 0;
    }
  };

  // ---- candidate wave: the stream
  const double* z_lane = nullptr; const double* lu_row = nullptr;
  bool fixed_l = false;
  double z_nx = 0.0, lu_nx = 0.0;
  // single-parameter schemes (round 5; R/kernel.R:66-133 plan_update_sequence: "ordered", an explicit sequence, "random"): ONE
  // parameter moves per step with the step's one variate -- the lane whose parameter it is takes dz, the others keep theirs.
  // (what a plan needs -- a lane's rank among the free parameters, their number, the first of them -- waits in LDS: three more values
  //  carried through the loop cost the JOINT scheme its registers, 7 % in the logistic latency form)
  const int scheme = A.scheme;
  int* const s_plan = reinterpret_cast<int*>(smem + LAT_PLAN);
  if constexpr (CANDS) {
    int zrank = 0, kfree = 0, first_free = 0;
    for (int j = k - 1; j >= 0; j--) { const int fr = A.fixed[j] ? 0 : 1; kfree += fr; if (j < jl) zrank += fr; if (fr) first_free = j; }
    s_plan[lane] = zrank;
    if (lane == 0) { s_plan[64] = kfree; s_plan[65] = first_free; }
    int zidx = zrank;
    if (zidx > kz - 1) zidx = kz > 0 ? kz - 1 : 0;   // lanes without a variate of their own read a valid neighbour (value unused)
    z_lane = A.fed_z + (cl * nsteps) * kz + zidx;
    lu_row = A.fed_logu + cl * nsteps;
    fixed_l = A.fixed[jl] != 0;
  }
  // (the plan of a step -- which lane moves -- is formed at the TOP of the step, outside the basic block in which candidates() rides
  //  beside the first chain's evaluation: its branches inside that block cost the joint scheme 6 % in the logistic latency form)
  bool keep_plan = false;
  auto plan = [&](int vn) {
    bool keep = fixed_l;                                // this lane's parameter does not move in this step
    if constexpr (ROLE == LAT_CANDS_S) {
      const long long ic = (long long)vn + 1 + A.step_off;            // the CALL's loop step of the proposal
      const int zrank = s_plan[lane];
      const int kfree = __builtin_amdgcn_readfirstlane(s_plan[64]), first_free = __builtin_amdgcn_readfirstlane(s_plan[65]);
      bool upd;
      if (scheme == FMCMC_SCHEME_ORDERED) {
        upd = !fixed_l && zrank == (int)((ic - 1) % kfree);
      } else if (scheme == FMCMC_SCHEME_EXPLICIT) {
        upd = jj == A.scheme_seq[(ic - 1) % A.scheme_len];
      } else {   // sample(which(!fixed), nsteps, TRUE)[i]; a single free parameter at position j makes R sample from 1:j
        const unsigned int pool = (kfree == 1) ? (unsigned int)(first_free + 1) : (unsigned int)kfree;
        const unsigned int idx = fmh_scheme_index(A.seed, (unsigned int)ic, (unsigned int)(A.chain_base + cl), pool);
        upd = (kfree == 1) ? (jj == (int)idx) : (!fixed_l && zrank == (int)idx);
        if (upd && act && A.scheme_cols && vn + 1 <= nsteps) A.scheme_cols[cl * A.nsteps_call + (ic - 1)] = jj;   // the plan, handed back
      }
      keep = !upd;
    }
    keep_plan = keep;
  };
  auto candidates = [&](int vn) {    // candidates of version vn + 1 and the log-uniform of decision vn -> buffer vn & 1
    const double dz = s_par[0 * 16 + jj] + s_par[1 * 16 + jj] * z_nx;
    double ca = th1 + dz, cr = th0 + dz;
    const bool keep = (ROLE == LAT_CANDS_S) ? keep_plan : fixed_l;
    if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) {
      if (act && !keep) {
        const double lb_l = s_par[2 * 16 + jj], ub_l = s_par[3 * 16 + jj];
        ca = reflect1(ca, lb_l, ub_l);
        cr = reflect1(cr, lb_l, ub_l);
      }
    }
    ca = keep ? th1 : ca;
    cr = keep ? th0 : cr;
    double* cb = s_cand + (vn & 1) * 128;
    cb[lane] = ca;
    cb[64 + lane] = cr;
    if (jj == 0) s_lu[(vn & 1) * LAT_ROWS + row] = lu_nx;
    // refills, consumed a whole step later (clamped rows: the tail re-reads)
    lu_nx = lu_row[vn < nsteps ? vn : nsteps - 1];
    z_nx = z_lane[(long long)(vn + 1 < nsteps ? vn + 1 : nsteps - 1) * kz];
  };
  // ---- the duties of waves 1 .. 3 in STAGES, run between the slot groups of the first chain's evaluation (stage g in front of group g:
  // the same basic block, in this order in the source -- left to itself the scheduler kept a duty's dependent chain in front of the
  // evaluation, ~450 cycles of latency that the barrier then waited for).  Branch-free: lanes without a result write to a sink.
  //   wave 1: n (log sigma + ln sqrt 2 pi) of theta1 (fmh_log_pn in its pieces: the same operations, the same bits)
  //   wave 2: both candidates of the next proposal and the log-uniform of the pending decision
  //   wave 3: sigma^2 and its reciprocal for div_finish (0: take the plain division)
  double dt_f = 0.0, dt_s = 0.0, dt_R = 0.0, dt_sg = 1.0;
  int dt_k = 0;
  bool dt_fast = false;
  const double* const s_tab = LG ? logit_table_align(smem + LAT_LDS_DOUBLES) : nullptr;
  const int nbl = A.intercept + A.p;
  const double hs_l = (LG && jj < nbl && A.lg_hs) ? A.lg_hs[jj] : 0.0;       // lane 16 c + j: the data-only sum of parameter j
  auto duty_stage = [&](int g, int vn) {     // vn: the version under evaluation (buffer vn & 1)
    if constexpr (LG) {
      if constexpr (ROLE == LAT_LOG) {
        if (g == 0) {   // sum_j b_j hs_j and sum_j b_j^2 of every row's chain (finish_logpost<LOGISTIC>: the same fma chains)
          double lin = 0.0, ss = 0.0;
          static_for<16>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if (j < nbl) {
              const double bj = row_bcast<j>(th1);
              lin = fmh_fma(bj, row_bcast<j>(hs_l), lin);
              ss = fmh_fma(bj, bj, ss);
            }
          });
          dt_f = lin; dt_s = ss;
        } else if (g == 1) {
          const double pri = (A.prior_div != 0.0) ? dt_s / A.prior_div : 0.0;
          const double v = (jj == 0) ? dt_f : pri;
          double* dst = (jj == 0) ? s_prep + (vn & 1) * 16 + 0 * LAT_ROWS + row : (jj == 2 ? s_prep + (vn & 1) * 16 + 2 * LAT_ROWS + row : s_sink + lane);
          *dst = v;
        }
      } else if constexpr (CANDS) {
        if (g == 0 && vn >= 2) candidates(vn);
      }
      return;
    }
    if constexpr (ROLE == LAT_LOG) {
      if (g == 0) {
        const double sigma = shfl_d(th1, (lane & 48) + k - 1);
        const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
        dt_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;                    // positive, finite, normal
        dt_sg = dt_fast ? sigma : 1.0;
        dt_f = fmh_log_split_pn_(dt_sg, &dt_k);
      } else if (g == 1) {
        dt_s = fmh_log_s_(dt_f);
      } else if (g == 2) {
        dt_R = fmh_log_R_(dt_s);
      } else if (g == 3) {
        const double t1 = fmh_log_fin_(dt_f, dt_k, 0.0, dt_s, dt_R) + FMH_K(FMH_LN_SQRT_2PI);   // = fmh_log(sigma) + ln sqrt 2 pi
        const double nt1 = dn * t1;
        const double v = (jj == 0) ? nt1 : (dt_fast ? 1.0 : 0.0);
        double* dst = (jj < 2) ? s_prep + (vn & 1) * 16 + jj * LAT_ROWS + row : s_sink + lane;
        *dst = v;
      }
    } else if constexpr (ROLE == LAT_RECIP) {
      if (g == 0) {
        const double sigma = shfl_d(th1, (lane & 48) + k - 1);
        const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
        dt_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;
        dt_sg = dt_fast ? sigma : 1.0;
      } else if (g == 1) {
        const double ss = dt_sg * dt_sg;
        const bool ok = dt_fast && mfr_div_safe(ss);
        const double rs = div_recip(ok ? ss : 1.0);
        const double v = (jj == 2) ? ss : (ok ? rs : 0.0);
        double* dst = (jj == 2 || jj == 3) ? s_prep + (vn & 1) * 16 + jj * LAT_ROWS + row : s_sink + lane;
        *dst = v;
      }
    } else if constexpr (CANDS) {
      if (g == 0 && vn >= 2) candidates(vn);   // (version 1's come from the prologue; behind the last step they go nowhere)
    }
  };
  auto logpost_of = [&](double tot, double sigma) -> double {
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) {
      f = fmh_nan();
    } else if (sigma == 0.0) {
      f = -fmh_inf();
    } else {
      double t1 = fmh_log(sigma) + FMH_K(FMH_LN_SQRT_2PI);
      double q = (0.5 * tot) / (sigma * sigma);
      f = -(dn * t1) - q;
    }
    if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
    return f;
  };
  // one chain's lane partial: the canonical fma chains of this lane's OPT observations, four slots in flight
  auto eval_chain = [&](int c, auto&& stage) -> double {
    const int l0 = 16 * c;
    double m00 = readlane_d(th1, l0);              // (a select, not a branch: the duty waves' preparation shares this basic block)
    m00 = ic ? m00 : 0.0;
    double b0[P > 0 ? P : 1];
#pragma unroll
    for (int j = 0; j < P; j++) b0[j] = readlane_d(th1, l0 + ic + j);
    if constexpr (LG) {      // (the table's argument is 64 eta: coefficients times 64, exact)
      m00 = m00 * FMH_LG_SCALE;
#pragma unroll
      for (int j = 0; j < P; j++) b0[j] = b0[j] * FMH_LG_SCALE;
    }
    // (uniform values kept in VGPRs: an fp64 fma with a scalar operand issues at 6 cycles where all-VGPR ones take 4.5)
    asm volatile("" : "+v"(m00));
#pragma unroll
    for (int j = 0; j < P; j++) asm volatile("" : "+v"(b0[j]));
    double a0 = 0.0;
    constexpr int G = 4, NGR = (OPT + G - 1) / G;
#pragma unroll
    for (int s0 = 0; s0 < OPT; s0 += G) {
      stage(s0 / G);
      double m[G];
#pragma unroll
      for (int u = 0; u < G; u++) m[u] = m00;
#pragma unroll
      for (int j = 0; j < P; j++)
#pragma unroll
        for (int u = 0; u < G; u++)
          if (s0 + u < OPT) m[u] = fmh_fma(xr[s0 + u][j], b0[j], m[u]);
      if constexpr (LG) {
        double us[G], gv[G];
#pragma unroll
        for (int u = 0; u < G; u++) us[u] = __builtin_fabs(m[u]);
        logit_g_vec<G, true>(us, gv, s_tab);
#pragma unroll
        for (int u = 0; u < G; u++) {
          const int s = s0 + u;
          if (s < OPT) {
            if (s == OPT - 1) a0 = a0 + gv[u] * wlast;       // (g * 1 == g; a slot without an observation adds +0)
            else if (s == OPT - 2) a0 = a0 + gv[u] * wprev;
            else a0 = a0 + gv[u];
          }
        }
      } else {
#pragma unroll
      for (int u = 0; u < G; u++)
        if (s0 + u < OPT) m[u] = yr[s0 + u] - m[u];
#pragma unroll
      for (int u = 0; u < G; u++) {
        const int s = s0 + u;
        if (s < OPT) {
          if (s == OPT - 1) a0 = fmh_fma(m[u] * wlast, m[u], a0);
          else if (s == OPT - 2) a0 = fmh_fma(m[u] * wprev, m[u], a0);   // (r * 1 == r: the same bits where the slot is full)
          else a0 = fmh_fma(m[u], m[u], a0);
        }
      }
      }
    }
#pragma unroll
    for (int g = NGR; g < 4; g++) stage(g);     // (fewer slot groups than stages: the rest behind the loop)
    return a0;
  };
  auto no_stage = [](int) {};

  // ---- prologue: what barrier 1 hands over
  if constexpr (CANDS) {
    z_nx = z_lane[(long long)(nsteps >= 2 ? 1 : 0) * kz];   // row 1: the variates of loop step 2
    lu_nx = lu_row[nsteps >= 2 ? 1 : 0];
    // (version 1 decides nothing: both candidates of version 2 are theta0 + dz; the refills inside fetch lu_row[1] -- the
    //  log-uniform of decision 2 -- and row 2)
    if constexpr (ROLE == LAT_CANDS_S) plan(1);
    candidates(1);
  }

  for (int v = 1; v <= nsteps; v++) {
    const int par = v & 1;
    if constexpr (ROLE == LAT_CANDS_S) plan(v);
    // ================= evaluation of version v of every chain, levels 1..32 of the tree =================
    // (the logarithm / the reciprocal of theta1's sigma ride in the same basic block as the first chain's evaluation)
    // (the tree of chain c - 1 runs in the same basic block as the evaluation of chain c: its six dependent levels hide
    //  under the next chain's fma chains; only the last chain's tree is exposed.  Lanes without a result write to a sink.)
    double a_prev = eval_chain(0, [&](int g) { duty_stage(g, v); });
    for (int c = 1; c < ncw; c++) {
      const double a_cur = eval_chain(c, no_stage);
      const double fs = wave_xor_sum(a_prev);
      double* dst = (lane == 0) ? s_fold + (par * LAT_ROWS + (c - 1)) * NW + wave : s_sink + lane;
      *dst = fs;
      a_prev = a_cur;
    }
    {
      const double fs = wave_xor_sum(a_prev);
      double* dst = (lane == 0) ? s_fold + (par * LAT_ROWS + (ncw - 1)) * NW + wave : s_sink + lane;
      *dst = fs;
    }
    lds_barrier();
    // ================= every wave: fold, closed form, accept, next proposal =================
    double wsum = s_fold[(par * LAT_ROWS + row) * NW + (lane & 7)];
    const double nt1 = s_prep[par * 16 + 0 * LAT_ROWS + row], sgf = s_prep[par * 16 + 1 * LAT_ROWS + row];
    const double ss = s_prep[par * 16 + 2 * LAT_ROWS + row], rs = s_prep[par * 16 + 3 * LAT_ROWS + row];
    const double lu = s_lu[par * LAT_ROWS + row];
    const double cand_a = s_cand[par * 128 + lane], cand_r = s_cand[par * 128 + 64 + lane];
    wsum = wsum + dpp_d<0xB1>(wsum);                          // level 64:  waves w, w ^ 1 (quad_perm [1,0,3,2])
    wsum = wsum + dpp_d<0x4E>(wsum);                          // level 128: quad_perm [2,3,0,1]
    const double tot = wsum + dpp_d<0x141>(wsum);             // level 256: row_half_mirror (quads are uniform)
    const double h = 0.5 * tot;
    double f1;
    if constexpr (LG) {          // nt1 = sum_j b_j hs_j, ss = the prior term (wave 1's duty)
      f1 = nt1 - tot;
      if (A.prior_div != 0.0) f1 = f1 - ss;
      if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
    } else {
      f1 = -nt1 - div_finish(h, ss, rs);
    }
    const double th1_eval = th1;
    const double ratio_f = f1 - f0;
    const bool rare = rowact && ((v == 1) || (status != FMCMC_CHAIN_OK) || (!LG && ((rs == 0.0) || !mfr_div_safe(h))) || fmh_isnan(ratio_f));
    bool keep_row = true, acc = false;
    if (__builtin_expect(!__any(rare), 1)) {
      acc = lu < ratio_f;
    } else {
      const double sigma = shfl_d(th1, (lane & 48) + k - 1);
      keep_row = false;
      if constexpr (LG) {
        (void)sigma;                                  // (f1 is what it is: the closed form has no special cases)
      } else if (sgf != 0.0) {
        f1 = -nt1 - h / ss;
        if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
      } else {
        f1 = logpost_of(tot, sigma);
      }
      if (v == 1) {
        f0 = f1;
        keep_row = true;
      } else if (status == FMCMC_CHAIN_OK) {
        const double ratio = f1 - f0;
        if (fmh_isnan(f1) || fmh_isnan(ratio)) {
          status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
          if constexpr (ROLE == LAT_KEEPER) {
            if (rowact) {
              if (jj == 0) { A.status[cl] = status; A.status_step[cl] = v + A.step_off; }
              if (jj < k) A.status_theta[cl * k + jj] = th1;
            }
          }
          flush_bits(v);
        } else {
          acc = lu < ratio;
          keep_row = true;
        }
      }
    }
    th0 = acc ? th1 : th0;
    f0 = acc ? f1 : f0;
    th1 = (status != FMCMC_CHAIN_OK) ? th1 : (acc ? cand_a : cand_r);   // (a failed chain keeps its theta1)
    // ---- the duties, in the shadow of the next evaluation (four different SIMDs)
    if constexpr (ROLE == LAT_KEEPER) {
      nacc += acc ? 1 : 0;
      bitword |= (acc ? 1u : 0u) << ((v - 1) & 31);
      if (keep_row && v > (int)A.burnin) {
        thin_ctr += 1;
        if (thin_ctr == (int)A.thin) {
          thin_ctr = 0;
          if (act) {
            *reinterpret_cast<double*>(s_ptr + srow8) = th0;
            if (d_ptr) *reinterpret_cast<double*>(d_ptr + srow8) = th1_eval;
          }
          if (l_ptr && rowact && jj == 0) *reinterpret_cast<double*>(l_ptr + srow8) = f1;
          srow8 += 8;
        }
      }
      if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
    }
  }
  if constexpr (ROLE == LAT_KEEPER) {
    if (rowact) {
      if (jj < k) A.theta0[cl * k + jj] = th0;
      if (jj == 0) {
        A.f0[cl] = f0;
        A.accept_count[cl] = nacc;   // (of THIS launch: launch_sweep adds the windows up)
        if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
      }
    }
  }
}

template <int KIND, int P, int OPT, int FAM = FMCMC_FAM_GAUSSIAN_LINREG>
__device__ __forceinline__ void lat_roles(const SweepArgs& A, double* smem) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave == 0) lat_steps<KIND, P, OPT, LAT_KEEPER, FAM>(A, smem);
  else if (wave == 1) lat_steps<KIND, P, OPT, LAT_LOG, FAM>(A, smem);
  else if (wave == 2) {   // (the scheme's plan in a loop of its own: inside the joint scheme's it cost that scheme registers -- 7 % of a step)
    if (A.scheme != FMCMC_SCHEME_JOINT) lat_steps<KIND, P, OPT, LAT_CANDS_S, FAM>(A, smem);
    else lat_steps<KIND, P, OPT, LAT_CANDS, FAM>(A, smem);
  }
  else if (wave == 3 && FAM != FMCMC_FAM_LOGISTIC) lat_steps<KIND, P, OPT, LAT_RECIP, FAM>(A, smem);
  else lat_steps<KIND, P, OPT, LAT_GENERIC, FAM>(A, smem);
}

// OPTMAX: the most observation slots a lane holds ((P + 1) OPTMAX doubles of x and y in VGPRs); the launch's (even) slot count
// A.spec_opt <= OPTMAX selects the step loop.
template <int KIND, int P, int OPTMAX, int FAM = FMCMC_FAM_GAUSSIAN_LINREG>
__global__ __launch_bounds__(NT) void mh_sweep_lat(const SweepArgs A) {
  extern __shared__ double smem[];
  if constexpr (FAM == FMCMC_FAM_LOGISTIC) logit_stage_table(logit_table_align(smem + LAT_LDS_DOUBLES));
  if (threadIdx.x < 64) {   // kernel constants of the candidate wave: mu, scale, lb, ub
    const int f = threadIdx.x >> 4, j = threadIdx.x & 15;
    const double* src = f == 0 ? A.mu : f == 1 ? A.scale : f == 2 ? A.lb : A.ub;
    smem[LAT_PAR + threadIdx.x] = (j < A.k) ? src[j] : 0.0;
  }
  __syncthreads();
  switch (A.spec_opt) {
#define LAT_CC(O_) case O_: if constexpr (O_ <= OPTMAX) lat_roles<KIND, P, O_, FAM>(A, smem); break;
    LAT_CC(2) LAT_CC(4) LAT_CC(6) LAT_CC(8) LAT_CC(10) LAT_CC(12) LAT_CC(14) LAT_CC(16) LAT_CC(18) LAT_CC(20)
#undef LAT_CC
    default: break;
  }
}

size_t lat_lds_bytes() { return sizeof(double) * (size_t)LAT_LDS_DOUBLES; }
size_t lat_logit_lds_bytes() { return sizeof(double) * (size_t)(LAT_LDS_DOUBLES + 2 + LG_LDS_DOUBLES); }

}  // namespace
