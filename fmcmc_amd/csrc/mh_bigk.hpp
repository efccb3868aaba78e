// mh_bigk.hpp -- mh_sweep_bigk: 64 < k <= 128 parameters (round 4; R/kernel_ram.R:93-121 and R/kernel_adapt.R:87-115 take any k,
// the authors' own benchmark is k = 100, playground/benchmarks.Rmd; until round 4 FMCMC_MAX_K was 64: the lane = row mapping
// of the owner wavefronts).
#pragma once

namespace {

// One chain per workgroup, thread r < kf = row r of the chain's matrices, every exchange through LDS behind workgroup barriers
// (lds_barrier): no wave-level primitive, so nothing here depends on a row fitting a lane.  The matrices are PACKED lower
// triangles (row r at r (r + 1) / 2): Sigma and its Cholesky factor (kernel_adapt) or the factor S (kernel_ram; the partial
// sums G of the product-form update are re-formed on the way down a row, as the oracle does, instead of being kept) --
// 2 x 66 KB at k = 128 where two squares would be 264 KB.  Same operations on every matrix element in the same order as the
// owner wavefronts of mh_sweep_kernel and the oracle (propose_adapt / propose_ram / ram_factor_update_canon), hence the same
// bits; the evaluation is the all-family eval_partials of the general kernel.  Built for completeness, not for speed: a
// Cholesky column costs two barriers (~0.3 us), a kernel_adapt step at k = 128 ~40 us on top of its evaluation.
// Takes: every family; kernel_normal(_reflective) / kernel_unif(_reflective) with the joint scheme, kernel_adapt (bw = 0,
// freq = 1), kernel_ram (any freq, bounds, constr).  The host refuses the rest for k > 64 with a message.
__host__ __device__ inline size_t bigk_lds_doubles(int k, int kf, int kind) {
  const size_t tri = (size_t)kf * (kf + 1) / 2;
  const size_t mats = (kind == FMCMC_KERNEL_ADAPT) ? 2 * tri : (kind == FMCMC_KERNEL_RAM ? tri : 0);
  return 5 * (size_t)k + (k / 2 + 1) + NW + 2 /* flags */ + (size_t)(k + 1) /* variates */ + (size_t)k /* scaled copy */ +
         2 * (size_t)k /* th0, th1 */ + 8 * (size_t)kf /* vectors */ + 2 * (size_t)kf /* scan buffers */ + mats + 4;
}

#ifdef FMH_WITH_BIGK_KERNEL   /* compiled into k_wide2.hip only; the engine takes bigk_lds_doubles above */
__global__ __launch_bounds__(NT) void mh_sweep_bigk(const SweepArgs A0) {
  SweepArgs A = A0;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, r = tid;
  const int k = A.k, kz = A.kz;
  double* s_mu = smem;
  double* s_scale = s_mu + k;
  double* s_lb = s_scale + k;
  double* s_ub = s_lb + k;
  double* s_hs = s_ub + k;
  int* s_which = (int*)(s_hs + k);
  double* s_part = s_hs + k + (k / 2 + 1);    // [NW]
  int* s_flag = (int*)(s_part + NW);          // [0] kf, [1] any-lane flag, [2] second flag
  double* s_z = s_part + NW + 2;              // [kz] variates of the step, [kz] the log accept uniform
  double* s_lgb = s_z + (k + 1);              // [k] scaled coefficients (logistic, eval_partials)
  double* th0 = s_lgb + k;                    // [k]
  double* th1 = th0 + k;                      // [k]
  __shared__ int s_kf_;
  if (tid == 0) {
    int kf0 = 0;
    for (int j = 0; j < k; j++)
      if (!A.fixed[j]) s_which[kf0++] = j;
    s_kf_ = kf0;
  }
  if (tid < k) {
    s_mu[tid] = A.mu[tid]; s_scale[tid] = A.scale[tid]; s_lb[tid] = A.lb[tid]; s_ub[tid] = A.ub[tid];
    s_hs[tid] = (A.family == FMCMC_FAM_LOGISTIC && A.lg_hs && tid < A.intercept + A.p) ? A.lg_hs[tid] : 0.0;
  }
  __syncthreads();
  const int kf = s_kf_;
  double* vv = th1 + k;        // [kf] x (adapt)
  double* vmp = vv + kf;       // [kf] mean_prev
  double* vmt = vmp + kf;      // [kf] mean_t
  double* vrs = vmt + kf;      // [kf] running sum of ans rows (adapt)
  double* vd = vrs + kf;       // [kf] d_j (ram)        | column broadcast (adapt: [0] the pivot)
  double* vk = vd + kf;        // [kf] kappa_j (ram)
  double* vP = vk + kf;        // [kf] prefix sums of z^2, inclusive
  double* vq = vP + kf;        // [kf] scan buffer A
  double* vt = vq + kf;        // [kf] scan buffer B
  double* vx = vt + kf;        // [kf] spare
  double* MA = vx + kf;        // packed lower: Sigma (adapt) / S (ram)
  const int tri = kf * (kf + 1) / 2;
  double* MB = MA + tri;       // packed lower: Cholesky factor (adapt)
  auto at = [](int i, int j) -> int { return i * (i + 1) / 2 + j; };   // j <= i

  const long long cl = blockIdx.x;            // one chain per workgroup
  const unsigned int cgid = (unsigned int)(A.chain_base + cl);
  const bool row = r < kf;                    // this thread owns matrix row r
  const bool par = r < k;                     // this thread owns parameter r
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const bool adapt = A.kind == FMCMC_KERNEL_ADAPT, ram = A.kind == FMCMC_KERNEL_RAM;

  // ---- chain state (uniform values are kept by every thread)
  double f0 = 0.0, f1 = 0.0;
  long long abs_iter = 0, nacc = 0;
  int have_mean = 0, nerr = 0, status = FMCMC_CHAIN_OK;
  unsigned int bitword = 0;
  if (par) { const double t = A.theta0[cl * k + r]; th0[r] = t; th1[r] = t; }
  if (adapt || ram) {
    for (int e = tid; e < tri; e += NT) {
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= e) i++;
      const int j = e - i * (i + 1) / 2;
      MA[e] = A.fresh ? ((i == j) ? 1.0 * A.eps : 0.0) : A.Sigma[(cl * kf + i) * kf + j];
      if (adapt) MB[e] = 0.0;
    }
    if (!A.fresh) {
      abs_iter = A.abs_iter[cl];
      if (A.nerrors) nerr = A.nerrors[cl];
      if (adapt) { have_mean = A.have_mean[cl]; if (row) vmp[r] = A.mean_prev[cl * kf + r]; }
    }
  }
  lds_barrier();

  double* thp[1] = {th1};
  unsigned sh_epoch = 0;
  auto evaluate = [&]() { eval_partials<1, 0, 0>(A, thp, s_part, nullptr, &sh_epoch, nullptr, s_lgb); };
  auto total = [&]() -> double {
    return ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) + ((s_part[4] + s_part[5]) + (s_part[6] + s_part[7]));
  };
  int thin_ctr = 0;
  long long srow = 0;
  auto store_row = [&](int i, double lpv) {
    if (i > burnin) {
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (par) {
          A.samples[(cl * k + r) * A.ldS + srow] = th0[r];
          if (A.draws) A.draws[(cl * k + r) * A.ldS + srow] = th1[r];
        }
        if (A.logpost && tid == 0) A.logpost[cl * A.ldS + srow] = lpv;
        srow += 1;
      }
    }
  };
  auto fail = [&](int i) {   // uniform: every thread knows `status`
    if (tid == 0) { A.status[cl] = status; A.status_step[cl] = i; }
    if (par) A.status_theta[cl * k + r] = th1[r];
  };

  // ---- row 1
  evaluate();
  lds_barrier();
  f0 = finish_logpost<0>(A, th1, total(), s_hs);
  f1 = f0;
  if (row) vrs[r] = th0[s_which[r]];
  store_row(1, f0);
  lds_barrier();

  for (int i = 2; i <= nsteps; i++) {
    // ================= variates of this step (canonical Philox stream, or the fed one) =================
    if (tid <= kz) {
      const unsigned int st = (unsigned int)(A.step_base + i);
      double v;
      if (tid == kz) v = (A.rng_mode == FMCMC_RNG_FED) ? A.fed_logu[cl * A.nsteps + (i - 1)] : fmh_log_accept_u(A.seed, st, cgid);
      else if (A.rng_mode == FMCMC_RNG_FED) v = A.fed_z[(cl * A.nsteps + (i - 1)) * kz + tid];
      else if (ram && A.ram_df > 0.0) v = fmh_student_t(A.seed, st, cgid, (unsigned int)tid, A.ram_df);
      else if (A.variate == 1) v = fmh_unif(A.seed, st, cgid, (unsigned int)tid);
      else v = fmh_normal(A.seed, st, cgid, (unsigned int)tid);
      s_z[tid] = v;
    }
    lds_barrier();
    bool ram_gate = false;
    // ================= proposal =================
    if (status == FMCMC_CHAIN_OK) {
      if (!adapt && !ram) {   // kernel_normal(_reflective), joint scheme (R/kernel_normal.R:67-72, :159-164)
        if (par) th1[r] = th0[r];
        lds_barrier();
        if (row) {
          const int j = s_which[r];
          double t = th0[j] + (s_mu[j] + s_scale[j] * s_z[r]);
          if (A.kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) t = reflect1(t, s_lb[j], s_ub[j]);
          th1[j] = t;
        }
      } else if (adapt) {     // R/kernel_adapt.R:117-180
        if (A.until > (double)abs_iter && abs_iter > A.warmup && i > 2) {
          const double t = (double)(abs_iter - 1);
          double x = 0, mp = 0, mt = 0;
          if (row) {
            x = th0[s_which[r]];
            mp = have_mean ? vmp[r] : (vrs[r] / (double)(i - 1));
            mt = (mp * t + x) / (t + 1);
            vv[r] = x; vmp[r] = mp; vmt[r] = mt;
          }
          lds_barrier();
          if (row) {
            const double c1 = (t - 1) / t, c2 = 1.0 / t;
            for (int b = 0; b <= r; b++) {      // (the element (r, b) of the symmetric update: the same bits as (b, r))
              const double ik = (b == r) ? 1.0 * A.eps : 0.0;
              const double inner = t * (mp * vmp[b]) - (t + 1) * (mt * vmt[b]) + x * vv[b] + 1e-5 * ik;
              MA[at(r, b)] = c1 * MA[at(r, b)] + c2 * inner;
            }
          }
          lds_barrier();
          if (row) vmp[r] = mt;
          have_mean = 1;
        }
        abs_iter += 1;
        // left-looking Cholesky, thread = row (twin of the oracle's chol_lower_canon)
        bool notpd = false;
        for (int j = 0; j < kf; j++) {
          double s = 0.0;
          if (row && r >= j) {
            s = MA[at(r, j)];
            for (int b = 0; b < j; b++) s = fmh_fma(-MB[at(r, b)], MB[at(j, b)], s);
            if (r == j) vd[0] = s;
          }
          lds_barrier();
          const double d = vd[0];
          if (!(d > 0.0) || !fmh_isfinite(d)) { notpd = true; break; }   // (uniform)
          const double ljj = fmh_sqrt(d);
          if (row && r == j) MB[at(j, j)] = ljj;
          else if (row && r > j) MB[at(r, j)] = s / ljj;
          lds_barrier();
        }
        if (notpd) {
          status = FMCMC_CHAIN_NOT_PD;
        } else {
          if (par) th1[r] = th0[r];
          lds_barrier();
          if (row) {
            double s = 0.0;
            for (int b = 0; b <= r; b++) s = fmh_fma(MB[at(r, b)], s_z[b], s);
            const int j = s_which[r];
            th1[j] = reflect1(th0[j] + (s_mu[j] + s), s_lb[j], s_ub[j]);
          }
        }
      } else {                // kernel_ram, R/kernel_ram.R:123-126 (theta1 keeps its previous values in fixed coordinates)
        if (row) {
          double s = 0.0;
          for (int b = r; b >= 0; b--) s = fmh_fma(MA[at(r, b)], s_z[b], s);
          const int j = s_which[r];
          th1[j] = th0[j] + s;
        }
        ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && (i % A.freq) == 0);
      }
      if (status != FMCMC_CHAIN_OK) fail(i);
    }
    lds_barrier();
    // ================= evaluation of f(theta1) =================
    evaluate();
    lds_barrier();
    // ================= kernel_ram: adaptation with f(theta1) of the un-reflected proposal (R/kernel_ram.R:129-152) =================
    double f1_pre = 0.0;
    bool have_f1 = false;
    if (ram) {
      if (status == FMCMC_CHAIN_OK) {
        if (ram_gate) {
          const double f1u = finish_logpost<0>(A, th1, total(), s_hs);
          f1_pre = f1u;
          have_f1 = !A.ram_bounded;
          double a_n = fmh_exp(f1u - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          double eta = (double)kf * fmh_exp(A.ram_neg_exp * fmh_log((double)i));
          if (eta > 1.0) eta = 1.0;
          // prefix sums of z^2: Hillis-Steele over the index, offsets 1, 2, 4, ... (the oracle's scan_sq_canon; the order a
          // wavefront's lane scan has for k <= 64)
          double* qa = vq;
          double* qb = vt;
          if (row) qa[r] = s_z[r] * s_z[r];
          lds_barrier();
          for (int s = 1; s < kf; s <<= 1) {
            if (row) qb[r] = (r >= s) ? qa[r] + qa[r - s] : qa[r];
            lds_barrier();
            double* tmp = qa; qa = qb; qb = tmp;
          }
          const double nrm2 = qa[kf - 1];
          const double cp = (eta * (a_n - A.arate)) / nrm2;
          if (tid == 0) s_flag[1] = 0;
          lds_barrier();
          if (cp != 0.0 && fmh_isfinite(cp)) {
            double dl = 0.0, kl = 0.0;
            if (row) {
              const double Pj1 = qa[r], Pj = (r == 0) ? 0.0 : qa[r - 1];
              const bool okl = ram_coef(cp, Pj, Pj1, s_z[r], dl, kl);
              if (!okl) s_flag[1] = 1;
              vd[r] = dl; vk[r] = kl;
            }
            lds_barrier();
            if (s_flag[1] != 0) {
              nerr += 1;
            } else if (row) {   // S'_rj = S_rj d_j + G_rj kappa_j, G re-formed from the diagonal down (ram_factor_update_canon)
              double G = 0.0;
              for (int j = r; j >= 0; j--) {
                const double sij = MA[at(r, j)];
                MA[at(r, j)] = fmh_fma(G, vk[j], sij * vd[j]);
                G = fmh_fma(sij, s_z[j], G);
              }
            }
          }
          if (A.constr) {  // Sigma <<- constr[which., which.] * Sigma (R/kernel_ram.R:149-150)
            if (row)
              for (int b = 0; b <= r; b++) MA[at(r, b)] = A.constr[r * kf + b] * MA[at(r, b)];
          }
          lds_barrier();
        }
        abs_iter += 1;
      }
      if (A.ram_bounded) {
        if (tid == 0) s_flag[2] = 0;
        lds_barrier();
        if (status == FMCMC_CHAIN_OK && row) {
          const int j = s_which[r];
          const double t0 = th1[j], t1 = reflect1(t0, s_lb[j], s_ub[j]);
          if (!(t1 == t0)) { th1[j] = t1; s_flag[2] = 1; }
        }
        lds_barrier();
        if (s_flag[2] != 0) {     // (uniform)
          evaluate();
          lds_barrier();
        }
      }
    }
    // ================= accept / store (R/mcmc.R:754-778) =================
    if (status == FMCMC_CHAIN_OK) {
      f1 = have_f1 ? f1_pre : finish_logpost<0>(A, th1, total(), s_hs);
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        fail(i);
      } else {
        const double lu = s_z[kz];
        if (lu < ratio) {
          if (par) th0[r] = th1[r];
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        lds_barrier();
        store_row(i, f1);
        if (adapt && row) vrs[r] = vrs[r] + th0[s_which[r]];
      }
    }
    if (A.accept_bits && tid == 0 && (((i - 1) & 31) == 31 || i == nsteps)) {
      A.accept_bits[cl * (long long)((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
      bitword = 0;
    }
    lds_barrier();
  }

  // ---- write state back
  if (par) A.theta0[cl * k + r] = th0[r];
  if (tid == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    if (adapt || ram) {
      A.abs_iter[cl] = abs_iter;
      if (A.nerrors) A.nerrors[cl] = nerr;
      if (adapt) A.have_mean[cl] = have_mean;
    }
  }
  if (adapt || ram) {
    for (int e = tid; e < kf * kf; e += NT) {
      const int a = e / kf, b = e % kf;
      // kernel_adapt hands on the full symmetric Sigma, kernel_ram its lower factor (+0 above the diagonal)
      A.Sigma[(cl * kf + a) * kf + b] = (b <= a) ? MA[at(a, b)] : (adapt ? MA[at(b, a)] : 0.0);
    }
    if (adapt && row) A.mean_prev[cl * kf + r] = vmp[r];
  }
}
#endif  // FMH_WITH_BIGK_KERNEL

}  // namespace
