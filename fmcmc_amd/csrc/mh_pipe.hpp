// mh_pipe.hpp -- rng_fill_kernel (canonical Philox stream -> HBM) and mh_sweep_pipe<P, OPT, KIND>, the software-pipelined
// VALU kernel (fallback of the wave-specialised one); scalar helpers shared with the kernels included after it.
#pragma once

namespace {

// ==============================================================================================
// Software-pipelined resident kernel (the headline path): kernel_normal / kernel_normal_reflective,
// Gaussian linear regression with P covariates held in VGPRs, 4 chains per workgroup in 2 groups.
//
//   half-step A(i): every wave evaluates group 0's proposals of step i      | owners of group 1 do
//                   (its 20 observations x 2 chains, data in registers)     | accept(i-1), propose(i)
//   barrier
//   half-step B(i): every wave evaluates group 1's proposals of step i      | owners of group 0 do
//                                                                           | accept(i), propose(i+1)
//   barrier
//
// The owner's scalar work (fold 512 lane partials, log sigma, two divisions, compare, stores, new
// proposal) is latency-bound; it sits in the SAME instruction stream as that wave's evaluation of the
// other group, so its stalls are filled with independent fp64 FMAs.  One barrier per half-step.
// Random variates come from HBM (rng_fill_kernel or host-fed), prefetched one step ahead into registers.
// ==============================================================================================
constexpr int PIPE_KMAX = 16; // parameters per chain supported by this kernel
constexpr int PIPE_TRS = 66;  // row stride (doubles) of the transposed lane-partial tile

// Canonical Philox stream materialised in HBM for the pipelined kernel (same layout as FED mode):
// logu[c][i-1] = log accept-uniform of loop step i, z[c][i-1][a] = a-th proposal variate of step i.
// Keeping Philox + AS241 (about 50 fp64 constants) out of the sweep kernel leaves its VGPR file
// to the observation data.  48 B per chain-step at k = 5: noise next to the 8 TB/s of HBM.
__global__ __launch_bounds__(256) void rng_fill_kernel(unsigned long long seed, long long step_base,
                                                       long long chain_base, long long nchains,
                                                       long long nsteps, int kz, double student_df,
                                                       double* __restrict__ logu, double* __restrict__ z) {
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= nchains * nsteps) return;
  const long long c = item / nsteps;
  const long long i = item - c * nsteps + 1;  // R's 1-based loop index
  const unsigned int st = (unsigned int)(step_base + i), cg = (unsigned int)(chain_base + c);
  if (i < 2) {  // row 1 draws nothing
    logu[item] = 0.0;
    for (int a = 0; a < kz; a++) z[item * kz + a] = 0.0;
    return;
  }
  logu[item] = fmh_log_accept_u(seed, st, cg);
  if (student_df > 0) {  // kernel_ram: qfun = rt(k, df)
    for (int a = 0; a < kz; a++) z[item * kz + a] = fmh_student_t(seed, st, cg, (unsigned int)a, student_df);
    return;
  }
  if (student_df < 0) {  // uniform kernels: the unif_rand() behind runif (R/kernel_unif.R:74)
    for (int a = 0; a < kz; a++) z[item * kz + a] = fmh_unif(seed, st, cg, (unsigned int)a);
    return;
  }
  for (int b = 0; 2 * b < kz; b++) {
    double u0, u1;
    fmh_uniform2(seed, st, cg, (unsigned int)b, FMH_STREAM_NORMAL, &u0, &u1);
    z[item * kz + 2 * b] = fmh_qnorm(u0);
    if (2 * b + 1 < kz) z[item * kz + 2 * b + 1] = fmh_qnorm(u1);
  }
}

__device__ __forceinline__ unsigned long long clk() {  // diagnostic stamp (debug mode 8 only)
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
// Register budget (512 threads -> 256 VGPRs): the x columns live in VGPRs (2*P*OPT = 120 at P=3,
// OPT=20), y lives in LDS (OPT*512*8 B = 80 KB, read back as b128 pairs), owner-only constants and
// addresses are kept in LDS / recomputed, so the steady-state loop runs without scratch traffic.
template <int P, int OPT, int KIND>
__global__ __launch_bounds__(NT) void mh_sweep_pipe(const SweepArgs A) {
  constexpr int CW = 4;
  static_assert(OPT % 2 == 0, "OPT must be even (y is read back in pairs)");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = A.k, kz = A.kz;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX] proposals read by the evaluation
  double* s_par = s_th1 + CW * PIPE_KMAX;          // [4][PIPE_KMAX]  mu, scale, lb, ub
  double* s_tr = s_par + 4 * PIPE_KMAX;            // [CW][8][PIPE_TRS] lane partials, transposed (below)
  double* s_y = s_tr + CW * 8 * PIPE_TRS;          // [OPT/2][NT][2] this workgroup's copy of y
  // Lane partial of canonical lane l goes to T[l & 7][l >> 3] (row stride PIPE_TRS = 66 doubles): the 64
  // b64 writes of a wavefront and the owner's 8 column reads T[j][q] are both bank-conflict free, whereas
  // the plain [lane] layout makes the fold (8 consecutive doubles per lane) a 16-way conflict.
  const int tr_slot = (tid & 7) * PIPE_TRS + (tid >> 3);
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int ic = A.intercept;

  // ---- this thread's observations (canonical lane = tid): x in registers, y in LDS
  double xr[OPT][P > 0 ? P : 1];
  double wlast;  // validity of the last slot (all earlier slots are full by dispatch: n > 512*(OPT-1))
  // y first (staged straight into LDS), x afterwards: keeps the prologue's peak register pressure below
  // the point where the allocator would spill long-lived values whose reloads land in the step loop
#pragma unroll
  for (int s = 0; s < OPT; s++) {
    const long long i = (long long)tid + (long long)NT * s;
    s_y[((s >> 1) * NT + tid) * 2 + (s & 1)] = (i < A.n) ? A.y[i] : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < OPT; s++) {
    const long long i = (long long)tid + (long long)NT * s;
    const bool valid = i < A.n;
#pragma unroll
    for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
    if (s == OPT - 1) wlast = valid ? 1.0 : 0.0;
  }
  if (tid < k) {
    s_par[0 * PIPE_KMAX + tid] = A.mu[tid];
    s_par[1 * PIPE_KMAX + tid] = A.scale[tid];
    s_par[2 * PIPE_KMAX + tid] = A.lb[tid];
    s_par[3 * PIPE_KMAX + tid] = A.ub[tid];
  }

  // ---- owner state: wave c owns chain c; lane j holds parameter j.  Everything wave-uniform is pinned
  // into SGPRs (readfirstlane) so that addresses are scalar and log-u arrives through a scalar load.
  const int myc = __builtin_amdgcn_readfirstlane(wave);
  const bool owner = (myc < ncw);
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + (owner ? myc : 0));  // local chain index
  const bool plane = owner && (lane < k);
  const int jl = (lane < k) ? lane : 0;
  const bool fixed_l = A.fixed[jl] != 0;
  int zidx = 0;  // index of this parameter among the free ones
  for (int j = 0; j < jl; j++) zidx += A.fixed[j] ? 0 : 1;
  double th0 = plane ? A.theta0[(long long)cl * k + lane] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0;
  unsigned int srow8 = 0;  // byte offset of the next kept row inside a column
  unsigned int bitword = 0;
  // 32-bit byte offsets off the SGPR base pointers (the dispatcher guarantees every array < 4 GiB)
  const unsigned int sd_off = (unsigned int)((((long long)cl * k + jl) * A.ldS) * 8);       // samples / draws column
  const unsigned int z_off = (unsigned int)((((long long)cl * nsteps) * kz + zidx) * 8);  // this lane's z column
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.ldS) * 8);
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;                        // scalar address
  const double dn = uniform_d((double)A.n);
  double z_nx = 0.0, lu_nx = 0.0;  // variates of the NEXT proposal / decision (prefetched)
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.fed_z) + (z_off + (unsigned int)row * (unsigned int)(kz * 8)));
  };

  if (tid < CW * PIPE_KMAX) s_th1[tid] = 0.0;
  lds_barrier();
  if (plane) s_th1[myc * PIPE_KMAX + lane] = th1;

  // ---- helpers -------------------------------------------------------------------------------
  // evaluation of one group (2 chains): lane partials -> s_tr
  auto eval_group = [&](int g) {
    const double* t0 = s_th1 + (2 * g) * PIPE_KMAX;
    const double* t1 = t0 + PIPE_KMAX;
    const double m00 = ic ? t0[0] : 0.0, m01 = ic ? t1[0] : 0.0;
    double b0[P > 0 ? P : 1], b1[P > 0 ? P : 1];
#pragma unroll
    for (int j = 0; j < P; j++) { b0[j] = t0[ic + j]; b1[j] = t1[ic + j]; }
    double a0 = 0.0, a1 = 0.0;
    const double2* yp = reinterpret_cast<const double2*>(s_y) + tid;
    double2 yy = yp[0];
#pragma unroll
    for (int s2 = 0; s2 < OPT / 2; s2++) {
      const double2 ynext = yp[(s2 + 1 < OPT / 2 ? s2 + 1 : s2) * NT];  // software prefetch of the next pair
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int s = 2 * s2 + h;
        const double yv = h ? yy.y : yy.x;
        double m0 = m00, m1 = m01;
#pragma unroll
        for (int j = 0; j < P; j++) { m0 = fmh_fma(xr[s][j], b0[j], m0); m1 = fmh_fma(xr[s][j], b1[j], m1); }
        const double r0 = yv - m0, r1 = yv - m1;
        if (s == OPT - 1) {
          a0 = fmh_fma(r0 * wlast, r0, a0);
          a1 = fmh_fma(r1 * wlast, r1, a1);
        } else {
          a0 = fmh_fma(r0, r0, a0);
          a1 = fmh_fma(r1, r1, a1);
        }
      }
      yy = ynext;
      // bound the scheduler's interleaving window: without it the unrolled loop is scheduled for maximal
      // ILP, the temporaries push the owner state into scratch and every store pays a memory round trip
      __builtin_amdgcn_sched_barrier(0);
    }
    s_tr[(2 * g) * (8 * PIPE_TRS) + tr_slot] = a0;
    s_tr[(2 * g + 1) * (8 * PIPE_TRS) + tr_slot] = a1;
  };
  // canonical tree over the 512 lane partials of this owner's chain
  auto fold_partials = [&]() -> double {
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;  // this lane folds canonical lanes 8*lane .. 8*lane+7
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    double v = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));  // levels 1,2,4
    return wave_xor_sum(v);                                                                         // levels 8..256
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
  auto store_row = [&](int r, double lpv) {
    if (r > burnin) {
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (plane) {
          *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sd_off + srow8)) = th0;
          if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sd_off + srow8)) = th1;
        }
        if (A.logpost && lane == 0) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = lpv;
        srow8 += 8;
      }
    }
  };
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0)
      A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };
  auto propose = [&](int inext) {  // proposal of loop step inext (uses z_nx, prefetches step inext + 1)
    if (plane) {
      double t = th0;
      if (!fixed_l) {
        t = th0 + (s_par[0 * PIPE_KMAX + lane] + s_par[1 * PIPE_KMAX + lane] * z_nx);
        if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) t = reflect1(t, s_par[2 * PIPE_KMAX + lane], s_par[3 * PIPE_KMAX + lane]);
        z_nx = ld_z(inext < nsteps ? inext : nsteps - 1);  // row (inext+1)-1, clamped: unconditional so the
                                                           // load lands in z_nx itself and is awaited a step later
      }
      th1 = t;
      s_th1[myc * PIPE_KMAX + lane] = t;
    }
  };
  auto accept = [&](int i) {  // decision of step i (partials of chain myc are in s_tr)
    const double tot = fold_partials();
    const double sigma = readlane_d(th1, k - 1);
    const double f1 = logpost_of(tot, sigma);
    const double ratio = f1 - f0;
    if (fmh_isnan(f1) || fmh_isnan(ratio)) {
      status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
      if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
      if (plane) A.status_theta[(long long)cl * k + lane] = th1;
      flush_bits(i);
      return;
    }
    const double lu = lu_nx;
    lu_nx = lu_row[i < nsteps ? i : nsteps - 1];  // log u of step i + 1 (clamped): scalar load, consumed a step later
    if (lu < ratio) {
      th0 = th1;
      f0 = uniform_d(f1);
      nacc += 1;
      bitword |= (1u << ((i - 1) & 31));
    }
    store_row(i, f1);
    if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
  };

  // ---- prologue: f0 of every chain, row 1, first proposals
  if (plane && !fixed_l && nsteps >= 2) z_nx = ld_z(1);
  if (owner && nsteps >= 2) lu_nx = lu_row[1];
  lds_barrier();  // s_th1, s_par, s_y are in place
  eval_group(0);
  eval_group(1);
  lds_barrier();
  if (owner) {
    const double tot = fold_partials();
    const double sigma = readlane_d(th1, k - 1);
    f0 = uniform_d(logpost_of(tot, sigma));
    store_row(1, f0);
    if ((myc >> 1) == 0) propose(2);   // group 1's first proposal is made in half-step A(2)
  }
  lds_barrier();

  // ---- steady state
  for (int i = 2; i <= nsteps; i++) {
    // ---------- half-step A(i): evaluate group 0 | group 1: accept(i-1), propose(i)
    if (owner && (myc >> 1) == 1 && status == FMCMC_CHAIN_OK && !(A.debug & 1)) {
      if (i > 2) accept(i - 1);
      if (status == FMCMC_CHAIN_OK) propose(i);
    }
    if (!(A.debug & 2)) eval_group(0);
    if (!(A.debug & 4)) lds_barrier();
    // ---------- half-step B(i): evaluate group 1 | group 0: accept(i), propose(i+1)
    if (owner && (myc >> 1) == 0 && status == FMCMC_CHAIN_OK && !(A.debug & 1)) {
      accept(i);
      if (status == FMCMC_CHAIN_OK && i < nsteps) propose(i + 1);
    }
    if (!(A.debug & 2)) eval_group(1);
    if (!(A.debug & 4)) lds_barrier();
  }
  // ---------- epilogue: group 1's last decision
  if (owner && (myc >> 1) == 1 && status == FMCMC_CHAIN_OK) accept(nsteps);

  // ---- write state back
  if (owner) {
    if (plane) A.theta0[(long long)cl * k + lane] = th0;
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    }
  }
}

}  // namespace
