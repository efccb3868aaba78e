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
// what the decision needs and the evaluation does not enter is prepared by DUTY waves on three different SIMDs while they
// evaluate, and handed over through LDS at the same barrier (double-buffered by the step's parity):
//     wave 1  the sigma-only half of the closed form of the proposal under evaluation: n (log sigma + ln sqrt 2 pi), sigma^2,
//             the reciprocal half of the division (div_recip / div_finish: bit for bit the quotient);
//     wave 2  BOTH candidates of the next proposal -- theta1 + dz (should this one be accepted) and theta0 + dz -- reflected
//             where the kernel reflects, and the log-uniform of the pending decision: the only wave that reads the stream;
//     wave 0  the keeper: rows of ans / draws / logpost, accept bitmap and counts, chain status, the state at the end.
// The decision itself is then ~25 instructions: fold, three fmas, a compare, selects.
// Same canonical lanes, same tree, same closed form as every other kernel: the bits do not depend on the form
// (tests/test_gpu_parity.py, test_latency_form_*).
// ==============================================================================================
constexpr int LAT_ROWS = 4;                      // 16-lane rows of a wavefront = chains a workgroup can hold (the dispatcher uses 1..3)
constexpr int LAT_FOLD = 0;                      // [2][LAT_ROWS][NW]   per-wave sums of every chain (levels 1..32 done)
constexpr int LAT_PREP = LAT_FOLD + 2 * LAT_ROWS * NW;   // [2][4][LAT_ROWS]  nt1, sigma^2, 1 / sigma^2, flag (0: general closed form, 1: sigma regular, 2: fast division)
constexpr int LAT_LU = LAT_PREP + 2 * 4 * LAT_ROWS;      // [2][LAT_ROWS]     log-uniform of the pending decision
constexpr int LAT_CAND = LAT_LU + 2 * LAT_ROWS;          // [2][2][64]        next proposal if accepted / if rejected, lane-wise
constexpr int LAT_LDS_DOUBLES = LAT_CAND + 2 * 2 * 64;

template <int KIND, int P, int OPT>
__device__ __forceinline__ void lat_steps(const SweepArgs& A, double* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz, ic = A.intercept;
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int cwl = A.spec_cw;
  const long long cg0 = (long long)blockIdx.x * cwl;
  const int ncw = (int)((A.nchains - cg0 < cwl) ? (A.nchains - cg0) : cwl);
  double* s_fold = smem + LAT_FOLD;
  double* s_prep = smem + LAT_PREP;
  double* s_lu = smem + LAT_LU;
  double* s_cand = smem + LAT_CAND;

  // ---- this lane's observations: canonical lane tid, slots s = 0 .. OPT - 1 (observation tid + 512 s)
  double xr[OPT][P > 0 ? P : 1], yr[OPT];
  double wlast = 1.0, wprev = 1.0;   // validity of the last two slots (OPT = the slot count rounded up to even)
#pragma unroll
  for (int s = 0; s < OPT; s++) {
    const long long i = (long long)tid + (long long)NT * s;
    const bool valid = i < A.n;
#pragma unroll
    for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
    yr[s] = valid ? A.y[i] : 0.0;
    if (s == OPT - 1) wlast = valid ? 1.0 : 0.0;
    if (s == OPT - 2) wprev = valid ? 1.0 : 0.0;
  }

  // ---- replicated chain state: lane 16 c + j <-> parameter j of chain c
  const int row = lane >> 4, jj = lane & 15;
  const bool act = row < ncw && jj < k;             // a parameter of a chain of this workgroup
  const bool rowact = row < ncw;
  const long long cl = cg0 + (rowact ? row : 0);    // (idle rows shadow chain 0 of the workgroup and never write)
  const int jl = (jj < k) ? jj : 0;
  const bool fixed_l = A.fixed[jl] != 0;
  double th0 = act ? A.theta0[cl * k + jj] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  int status = (A.win_cont && rowact) ? A.status[cl] : FMCMC_CHAIN_OK;
  const double dn = (double)A.n;
  const bool keeper = wave == 0, prepw = wave == 1, candw = wave == 2;

  // ---- keeper (wave 0): rows, bitmap, counts
  int nacc = 0, thin_ctr = A.thin_ctr0;
  unsigned int bitword = 0;
  char* const s_ptr = reinterpret_cast<char*>(A.samples) + ((cl * k + jl) * A.ldS) * 8;
  char* const d_ptr = A.draws ? reinterpret_cast<char*>(A.draws) + ((cl * k + jl) * A.ldS) * 8 : nullptr;
  char* const l_ptr = A.logpost ? reinterpret_cast<char*>(A.logpost) + (cl * A.ldS) * 8 : nullptr;
  unsigned int srow8 = 0;
  auto flush_bits = [&](int i) {   // (bits_stride: words per chain of the whole call's bitmap, set by launch_sweep for every launch)
    if (keeper && A.accept_bits && rowact && jj == 0) {
      unsigned int* w = A.accept_bits + (cl * A.bits_stride + ((i - 1) >> 5));
      // the first word of a continuation window also holds the last bit of the window before it
      *w = (A.win_cont && i <= 32) ? (*w | bitword) : bitword;
    }
    bitword = 0;
  };

  // ---- candidate wave (wave 2): the stream
  int zidx = 0;
  for (int j = 0; j < jl; j++) zidx += A.fixed[j] ? 0 : 1;
  if (zidx > kz - 1) zidx = kz > 0 ? kz - 1 : 0;   // lanes without a variate of their own read a valid neighbour (value unused)
  const double* const z_lane = A.fed_z + (cl * nsteps) * kz + zidx;
  const double* const lu_row = A.fed_logu + cl * nsteps;
  const double mu_l = A.mu[jl], sc_l = A.scale[jl], lb_l = A.lb[jl], ub_l = A.ub[jl];
  auto ld_z = [&](int r) -> double { return z_lane[(long long)r * kz]; };
  double z_nx = 0.0, lu_nx = 0.0;
  auto candidates = [&](int vn) {    // candidates of version vn + 1 and the log-uniform of decision vn -> buffer vn & 1
    const double dz = mu_l + sc_l * z_nx;
    double ca = th1 + dz, cr = th0 + dz;
    if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) {
      if (act && !fixed_l) { ca = reflect1(ca, lb_l, ub_l); cr = reflect1(cr, lb_l, ub_l); }
    }
    ca = fixed_l ? th1 : ca;
    cr = fixed_l ? th0 : cr;
    double* cb = s_cand + (vn & 1) * 128;
    cb[lane] = ca;
    cb[64 + lane] = cr;
    if (jj == 0) s_lu[(vn & 1) * LAT_ROWS + row] = lu_nx;
    // refills, consumed a whole step later (clamped rows: the tail re-reads)
    lu_nx = lu_row[vn < nsteps ? vn : nsteps - 1];
    z_nx = ld_z(vn + 1 < nsteps ? vn + 1 : nsteps - 1);
  };
  // ---- preparation wave (wave 1): the sigma-only half of the closed form of theta1
  auto prepare = [&](int vn) {       // -> buffer vn & 1
    const double sigma = shfl_d(th1, (lane & 48) + k - 1);
    const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
    const bool sg_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;            // positive, finite, normal
    const double sg = sg_fast ? sigma : 1.0;
    const double t1 = fmh_log_pn(sg) + FMH_K(FMH_LN_SQRT_2PI);           // same bits as fmh_log(sigma) on this range
    const double nt1 = dn * t1, ss = sg * sg;
    const bool ok = sg_fast && mfr_div_safe(ss);
    const double rs = div_recip(ok ? ss : 1.0);
    if (jj < 4) {
      const double v = jj == 0 ? nt1 : jj == 1 ? ss : jj == 2 ? rs : (ok ? 2.0 : (sg_fast ? 1.0 : 0.0));
      s_prep[(vn & 1) * 16 + jj * LAT_ROWS + row] = v;
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

  // ---- prologue: what barrier 1 hands over
  if (candw) {
    z_nx = ld_z(nsteps >= 2 ? 1 : 0);          // row 1: the variates of loop step 2
    lu_nx = lu_row[nsteps >= 2 ? 1 : 0];
    // (version 1 decides nothing: both candidates of version 2 are theta0 + dz; candidates() refills with lu_row[1] and row 2)
    const double lu_keep = lu_nx;
    candidates(1);
    lu_nx = lu_keep;                           // decision 2 takes lu_row[1]
  }
  if (prepw) prepare(1);

  bool st_keep = false, st_acc = false;        // row of the step just decided, stored by the keeper behind the decision
  double st_th0 = 0.0, st_th1 = 0.0, st_f1 = 0.0;
  for (int v = 1; v <= nsteps; v++) {
    const int par = v & 1;
    // ================= evaluation of version v of every chain, levels 1..32 of the tree =================
    for (int c = 0; c < ncw; c++) {
      const int l0 = 16 * c;
      double m00 = ic ? readlane_d(th1, l0) : 0.0;
      double b0[P > 0 ? P : 1];
#pragma unroll
      for (int j = 0; j < P; j++) b0[j] = readlane_d(th1, l0 + ic + j);
      // (uniform values kept in VGPRs: an fp64 fma with a scalar operand issues at 6 cycles where all-VGPR ones take 4.5)
      asm volatile("" : "+v"(m00));
#pragma unroll
      for (int j = 0; j < P; j++) asm volatile("" : "+v"(b0[j]));
      double a0 = 0.0;
#pragma unroll
      for (int s = 0; s < OPT; s++) {
        double m0 = m00;
#pragma unroll
        for (int j = 0; j < P; j++) m0 = fmh_fma(xr[s][j], b0[j], m0);
        const double r0 = yr[s] - m0;
        if (s == OPT - 1) a0 = fmh_fma(r0 * wlast, r0, a0);
        else if (s == OPT - 2) a0 = fmh_fma(r0 * wprev, r0, a0);   // (r0 * 1 == r0: the same bits where the slot is full)
        else a0 = fmh_fma(r0, r0, a0);
      }
      const double fs = wave_xor_sum(a0);
      if (lane == 0) s_fold[(par * LAT_ROWS + c) * NW + wave] = fs;
    }
    lds_barrier();
    // ================= every wave: fold, closed form, accept, next proposal =================
    double wsum = s_fold[(par * LAT_ROWS + row) * NW + (lane & 7)];
    const double nt1 = s_prep[par * 16 + 0 * LAT_ROWS + row], ss = s_prep[par * 16 + 1 * LAT_ROWS + row];
    const double rs = s_prep[par * 16 + 2 * LAT_ROWS + row], flag = s_prep[par * 16 + 3 * LAT_ROWS + row];
    const double lu = s_lu[par * LAT_ROWS + row];
    const double cand_a = s_cand[par * 128 + lane], cand_r = s_cand[par * 128 + 64 + lane];
    wsum = wsum + dpp_d<0xB1>(wsum);                          // level 64:  waves w, w ^ 1 (quad_perm [1,0,3,2])
    wsum = wsum + dpp_d<0x4E>(wsum);                          // level 128: quad_perm [2,3,0,1]
    const double tot = wsum + dpp_d<0x141>(wsum);             // level 256: row_half_mirror (quads are uniform)
    const double h = 0.5 * tot;
    double f1 = -nt1 - div_finish(h, ss, rs);
    const double th1_eval = th1;
    const double ratio_f = f1 - f0;
    const bool rare = rowact && ((v == 1) || (status != FMCMC_CHAIN_OK) || (flag != 2.0) || !mfr_div_safe(h) || fmh_isnan(ratio_f));
    bool keep_row = true, acc = false;
    if (__builtin_expect(!__any(rare), 1)) {
      acc = lu < ratio_f;
    } else {
      const double sigma = shfl_d(th1, (lane & 48) + k - 1);
      keep_row = false;
      if (flag != 0.0) {
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
          if (keeper && rowact) {
            if (jj == 0) { A.status[cl] = status; A.status_step[cl] = v + A.step_off; }
            if (jj < k) A.status_theta[cl * k + jj] = th1;
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
    // ---- the duties, in the shadow of the next evaluation (three different SIMDs)
    if (v < nsteps) {
      if (candw) candidates(v + 1);
      if (prepw) prepare(v + 1);
    }
    if (keeper) {
      st_keep = keep_row; st_acc = acc; st_th0 = th0; st_th1 = th1_eval; st_f1 = f1;
      nacc += st_acc ? 1 : 0;
      bitword |= (st_acc ? 1u : 0u) << ((v - 1) & 31);
      if (st_keep && v > burnin) {
        thin_ctr += 1;
        if (thin_ctr == thin) {
          thin_ctr = 0;
          if (act) {
            *reinterpret_cast<double*>(s_ptr + srow8) = st_th0;
            if (d_ptr) *reinterpret_cast<double*>(d_ptr + srow8) = st_th1;
          }
          if (l_ptr && rowact && jj == 0) *reinterpret_cast<double*>(l_ptr + srow8) = st_f1;
          srow8 += 8;
        }
      }
      if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
    }
  }
  if (keeper && rowact) {
    if (jj < k) A.theta0[cl * k + jj] = th0;
    if (jj == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;   // (of THIS launch: launch_sweep adds the windows up)
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    }
  }
}

// OPTMAX: the most observation slots a lane holds ((P + 1) OPTMAX doubles of x and y in VGPRs); the launch's (even) slot count
// A.spec_opt <= OPTMAX selects the step loop.
template <int KIND, int P, int OPTMAX>
__global__ __launch_bounds__(NT) void mh_sweep_lat(const SweepArgs A) {
  extern __shared__ double smem[];
  switch (A.spec_opt) {
#define LAT_CC(O_) case O_: if constexpr (O_ <= OPTMAX) lat_steps<KIND, P, O_>(A, smem); break;
    LAT_CC(2) LAT_CC(4) LAT_CC(6) LAT_CC(8) LAT_CC(10) LAT_CC(12) LAT_CC(14) LAT_CC(16) LAT_CC(18) LAT_CC(20)
#undef LAT_CC
    default: break;
  }
}

size_t lat_lds_bytes() { return sizeof(double) * (size_t)LAT_LDS_DOUBLES; }

}  // namespace
