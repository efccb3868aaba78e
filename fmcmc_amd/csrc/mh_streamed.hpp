// mh_streamed.hpp -- mh_sweep_kernel<CW, P, OPT, KIND>: the general kernel (every family, kernel, scheme and k <= 64; data
// streamed from L2 or, for two shapes, resident in VGPRs) with in-kernel Philox tiles.
#pragma once

namespace {

// ---- the sweep kernel ----------------------------------------------------------------------
// P < 0 : streamed evaluation (any family, any n, p: data re-read from L2 every step)
// P >= 0: register-resident Gaussian linear regression with P covariates: each thread keeps its
//         OPT observations (x[P], y) in VGPRs for the whole sweep; n in (512*(OPT-4), 512*OPT].
constexpr int RES_MASKED = 4;  // trailing observation slots that carry a validity mask

// KIND > 0 compiles exactly one proposal kernel in (resident variants); KIND == 0 keeps all four
// behind the runtime A.kind (streamed variants).
// FAM > 0 compiles one model family in, MINB is the number of workgroups per CU the register allocation must allow
// (the logistic model is bound by fp64 instruction issue: with 128 VGPRs two workgroups share a CU, 4 waves per SIMD).
// kernel_ram on the owner wave, lane = row of the lower factor S (LDS, row stride LD, +0 above the diagonal).
// ram_propose_rows: (S z)_i as an fma chain from the last column down to column 0 (the entries above the diagonal add +0
// exactly), keeping every partial sum G_ij = sum_{m = j+1..i} S_im z_m: they are what the factor update needs, which then
// is one mul + fma per element with nothing carried (ram_update_rows; d_j, kappa_j in dk[0..kf), dk[kf..2 kf), from
// ram_coef).  REAL functions on purpose: inlined into the sweep kernels the loops share their register allocation, and the
// instantiations that sit at 256 VGPRs spill inside them.
typedef __attribute__((address_space(3))) double* lds_dptr_t;
// (both loops: ONE exec region around everything, loads unconditional and a group of four columns ahead of the arithmetic --
//  with a predicate per store the compiler sank the loads into a branch per column: one LDS round trip per column, 4 us
//  per update at k = 50)
__device__ __attribute__((noinline)) double ram_propose_rows(lds_dptr_t S, lds_dptr_t G, lds_dptr_t z, int LD_, int kf_) {
  const int lane = threadIdx.x & 63;
  const int LD = __builtin_amdgcn_readfirstlane(LD_), kf = __builtin_amdgcn_readfirstlane(kf_);
  double s = 0.0;
  if (lane < kf) {
    const lds_dptr_t row = S + lane * LD, grow = G + lane * LD;
    int j = kf - 1;
    double sc[4], zc[4];
    if (j >= 3) {
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = row[j - u]; zc[u] = z[j - u]; }
    }
    for (; j >= 3; j -= 4) {
      const int jn = (j - 4 >= 3) ? j - 4 : j;       // the next group (the last round re-reads its own: unused)
      double sn[4], zn[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { sn[u] = row[jn - u]; zn[u] = z[jn - u]; }
      const double g0 = s, g1 = fmh_fma(sc[0], zc[0], g0), g2 = fmh_fma(sc[1], zc[1], g1), g3 = fmh_fma(sc[2], zc[2], g2);
      s = fmh_fma(sc[3], zc[3], g3);
      grow[j] = g0; grow[j - 1] = g1; grow[j - 2] = g2; grow[j - 3] = g3;
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = sn[u]; zc[u] = zn[u]; }
    }
    for (; j >= 0; j--) {
      const double sij = row[j];
      grow[j] = s;
      s = fmh_fma(sij, z[j], s);
    }
  }
  return s;
}
__device__ __attribute__((noinline)) void ram_update_rows(lds_dptr_t S, lds_dptr_t G, lds_dptr_t dk, int LD_, int kf_) {
  const int lane = threadIdx.x & 63;
  const int LD = __builtin_amdgcn_readfirstlane(LD_), kf = __builtin_amdgcn_readfirstlane(kf_);
  if (lane < kf) {
    const lds_dptr_t row = S + lane * LD, grow = G + lane * LD, kap = dk + kf;
    const int nq = kf & ~3;
    double sc[4], gc[4], dc[4], kc[4];
    if (nq > 0) {
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = row[u]; gc[u] = grow[u]; dc[u] = dk[u]; kc[u] = kap[u]; }
    }
    int j = 0;
    for (; j < nq; j += 4) {
      const int jn = (j + 4 < nq) ? j + 4 : j;
      double sn[4], gn[4], dn[4], kn[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { sn[u] = row[jn + u]; gn[u] = grow[jn + u]; dn[u] = dk[jn + u]; kn[u] = kap[jn + u]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double nw = fmh_fma(gc[u], kc[u], sc[u] * dc[u]);
        row[j + u] = (lane >= j + u) ? nw : sc[u];        // (above the diagonal: the +0 it holds)
      }
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = sn[u]; gc[u] = gn[u]; dc[u] = dn[u]; kc[u] = kn[u]; }
    }
    for (; j < kf; j++) {
      const double sij = row[j], nw = fmh_fma(grow[j], kap[j], sij * dk[j]);
      row[j] = (lane >= j) ? nw : sij;
    }
  }
}

// Both at once, for an owner that goes straight from the adaptation of step i to the proposal of step i + 1 (mh_wide2.hpp):
// S'_ij = fma(G_ij, kappa_j, S_ij d_j) is used the moment it exists -- (S' z')_i accumulated from the last column down, its
// partial sums G'_ij written over G_ij.  One pass over the two matrices instead of two; the same operations on every
// element in the same order as ram_update_rows followed by ram_propose_rows.
__device__ __attribute__((noinline)) double ram_update_propose_rows(lds_dptr_t S, lds_dptr_t G, lds_dptr_t dk, lds_dptr_t z,
                                                                    int LD_, int kf_) {
  const int lane = threadIdx.x & 63;
  const int LD = __builtin_amdgcn_readfirstlane(LD_), kf = __builtin_amdgcn_readfirstlane(kf_);
  double s = 0.0;
  if (lane < kf) {
    const lds_dptr_t row = S + lane * LD, grow = G + lane * LD, kap = dk + kf;
    int j = kf - 1;
    double sc[4], gc[4], dc[4], kc[4], zc[4];
    if (j >= 3) {
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = row[j - u]; gc[u] = grow[j - u]; dc[u] = dk[j - u]; kc[u] = kap[j - u]; zc[u] = z[j - u]; }
    }
    for (; j >= 3; j -= 4) {
      const int jn = (j - 4 >= 3) ? j - 4 : j;       // the next group (the last round re-reads its own before it is rewritten: unused)
      double sn[4], gn[4], dn[4], kn[4], zn[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { sn[u] = row[jn - u]; gn[u] = grow[jn - u]; dn[u] = dk[jn - u]; kn[u] = kap[jn - u]; zn[u] = z[jn - u]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double nw = (lane >= j - u) ? fmh_fma(gc[u], kc[u], sc[u] * dc[u]) : sc[u];
        row[j - u] = nw;
        grow[j - u] = s;
        s = fmh_fma(nw, zc[u], s);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = sn[u]; gc[u] = gn[u]; dc[u] = dn[u]; kc[u] = kn[u]; zc[u] = zn[u]; }
    }
    for (; j >= 0; j--) {
      const double sij = row[j];
      const double nw = (lane >= j) ? fmh_fma(grow[j], kap[j], sij * dk[j]) : sij;
      row[j] = nw;
      grow[j] = s;
      s = fmh_fma(nw, z[j], s);
    }
  }
  return s;
}

// FEDONLY: an instantiation for calls whose variates come from a materialised stream (the observation-sharded logistic sweep since
// round 5): no Philox, no AS241 with its ~50 constants in the kernel body -- they were what its step loop kept in scratch.
template <int CW, int P, int OPT, int KIND, int FAM = 0, int MINB = 1, bool FEDONLY = false>
__global__ __launch_bounds__(NT, MINB) void mh_sweep_kernel(const SweepArgs A0) {
  constexpr bool RESIDENT = (P >= 0);
  SweepArgs A = A0;
  const bool fed = FEDONLY ? true : (A0.rng_mode == FMCMC_RNG_FED);
  if constexpr (KIND > 0) A.kind = KIND;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = A.k;
  const int TB = A.tb, kz = A.kz;
  // ---- shared layout: kernel parameters, which[], partials, RNG tile, then CW chain blocks
  double* s_mu = smem;
  double* s_scale = s_mu + k;
  double* s_lb = s_scale + k;
  double* s_ub = s_lb + k;
  double* s_hs = s_ub + k;                  // [k] logistic: the data-only sums of the linear part (SweepArgs.lg_hs)
  int* s_which = (int*)(s_hs + k);          // [k] ints (k/2+1 doubles)
  double* s_part = s_hs + k + (k / 2 + 1);  // [NW*CW]
  int* s_flag = (int*)(s_part + NW * CW);   // [2] ints
  double* s_zt = s_part + NW * CW + 1;      // [CW][TB][kz] proposal variates of the tile
  double* s_lu = s_zt + CW * TB * kz;       // [CW][TB]     log accept uniforms of the tile
  double* s_lgb = s_lu + CW * TB;           // [CW][k]      logistic: the coefficient vectors times 64 (eval_partials)
  double* s_tr = s_lgb + CW * k;            // RESIDENT: [CW][NT] lane partials
  double* s_chains = s_tr + (RESIDENT ? CW * NT : 0);   // [CW][CHS], then (logistic-only instantiations) the g table

  __shared__ int s_kf;
  if (tid == 0) {
    int kf = 0;
    for (int j = 0; j < k; j++)
      if (!A.fixed[j]) s_which[kf++] = j;
    s_kf = kf;
    s_flag[0] = 0;
  }
  if (tid < k) {
    s_mu[tid] = A.mu[tid];
    s_scale[tid] = A.scale[tid];
    s_lb[tid] = A.lb[tid];
    s_ub[tid] = A.ub[tid];
    s_hs[tid] = (A.family == FMCMC_FAM_LOGISTIC && A.lg_hs && tid < A.intercept + A.p) ? A.lg_hs[tid] : 0.0;
  }
  __syncthreads();
  const int kf = s_kf;
  const int LD = kf | 1;
  const int CHS = chain_lds_doubles(k, kf, A.kind);
  const double* s_sptab = nullptr;
  if constexpr (FAM == FMCMC_FAM_LOGISTIC) {
    double* tabs = logit_table_align(s_chains + CW * CHS);    // 16-byte aligned pairs
    logit_stage_table(tabs);
    if constexpr (P < 0 && OPT == 2) logit_reset_turns(tabs);     // (the observation-sharded form: logit_shard's control words)
    s_sptab = tabs;
  }
  if constexpr (FAM == FMCMC_FAM_GAUSSIAN_LINREG && P < 0 && OPT > 0) {
    if (A.sh_mfma) {   // observation-sharded evaluation on the matrix cores: this workgroup's slice in operand layout
      double* blk = s_chains + CW * CHS + ((CW * CHS) & 1);
      const double* src = A.sh_mfma + (long long)blockIdx.x * A.sh_mblk;
      for (int i = tid; i < A.sh_mblk; i += NT) blk[i] = src[i];
      s_sptab = blk;
    } else if (A.sh_long) {   // long-data form: the block holds the residuals of a group of chains (shard_long), nothing to stage
      s_sptab = s_chains + CW * CHS + ((CW * CHS) & 1);
    }
  }
  const long long cg0 = (long long)blockIdx.x * CW;  // first local chain of this workgroup
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const bool adaptive = (A.kind == FMCMC_KERNEL_ADAPT || A.kind == FMCMC_KERNEL_RAM);

  // owner wavefront of chain c is wave c (CW <= NW)
  const int myc = wave;                 // chain slot owned by this wavefront
  const bool owner = (myc < ncw);
  const long long cl = cg0 + myc;       // local chain index
  const unsigned int cgid = (unsigned int)(A.chain_base + cl);
  ChainLds L = chain_lds(s_chains + (owner ? myc : 0) * CHS, k, kf, A.kind);

  double* thp[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) thp[c] = s_chains + (c < ncw ? c : 0) * CHS + k;  // th1 of chain c

  // ---- RESIDENT: this thread's observations live in registers for the whole sweep
  constexpr int PR = RESIDENT ? (P > 0 ? P : 1) : 1;
  constexpr int OR = RESIDENT ? OPT : 1;
  double xr[OR][PR];
  double yr[OR];
  double wm[RES_MASKED];
  if constexpr (RESIDENT) {
#pragma unroll
    for (int s = 0; s < OPT; s++) {
      const long long i = (long long)tid + (long long)NT * s;
      const bool valid = i < A.n;
      yr[s] = valid ? A.y[i] : 0.0;
#pragma unroll
      for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
      if (s >= OPT - RES_MASKED) wm[s - (OPT - RES_MASKED)] = valid ? 1.0 : 0.0;
    }
  }

  unsigned sh_epoch = 0;   // grid-barrier epoch of the observation-sharded evaluation (uniform over the grid)
#ifdef FMCMC_STAMP
  Stamps stamps; for (int q = 0; q < 16; q++) stamps.acc[q] = 0;
  stamps.prev = 0;
  Stamps* const stp = &stamps;
#else
  Stamps* const stp = nullptr;
#endif
  // collective evaluation of f(theta1) for all chains of the workgroup; on return s_part holds
  // what finish needs (streamed: 8 wave partials per chain; resident: 2 half totals per chain)
  auto evaluate = [&]() {
    if constexpr (RESIDENT) {
      double m0[CW], bb[CW][PR], acc[CW];
#pragma unroll
      for (int c = 0; c < CW; c++) {
        m0[c] = A.intercept ? thp[c][0] : 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) bb[c][j] = thp[c][A.intercept + j];
        acc[c] = 0.0;
      }
#pragma unroll
      for (int s = 0; s < OPT; s++) {
#pragma unroll
        for (int c = 0; c < CW; c++) {
          double m = m0[c];
#pragma unroll
          for (int j = 0; j < P; j++) m = fmh_fma(xr[s][j], bb[c][j], m);
          double r = yr[s] - m;
          if (s >= OPT - RES_MASKED) acc[c] = fmh_fma(r * wm[s - (OPT - RES_MASKED)], r, acc[c]);
          else acc[c] = fmh_fma(r, r, acc[c]);
        }
      }
      // canonical tree through LDS: lane partials -> [chain][lane]; wave (c, h) folds 256 lanes
#pragma unroll
      for (int c = 0; c < CW; c++) s_tr[c * NT + tid] = acc[c];
      __syncthreads();
      for (int job = wave; job < 2 * CW; job += NW) {
        const int c = job % CW, h = job / CW;
        const double* src = s_tr + c * NT + 256 * h + 4 * lane;
        double v = (src[0] + src[1]) + (src[2] + src[3]);      // levels 1, 2
        v = wave_xor_sum(v);                                   // levels 4..128
        if (lane == 0) s_part[h * CW + c] = v;
      }
    } else {
      eval_partials<CW, FAM, (P < 0 ? OPT : 0)>(A, thp, s_part, s_sptab, &sh_epoch, stp, s_lgb);
    }
  };
  auto total_of = [&](int c) -> double {
    if constexpr (RESIDENT) {
      return s_part[0 * CW + c] + s_part[1 * CW + c];           // level 256
    } else {
      double w0 = s_part[0 * CW + c], w1 = s_part[1 * CW + c], w2 = s_part[2 * CW + c], w3 = s_part[3 * CW + c];
      double w4 = s_part[4 * CW + c], w5 = s_part[5 * CW + c], w6 = s_part[6 * CW + c], w7 = s_part[7 * CW + c];
      return ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7));  // levels 64, 128, 256
    }
  };

  // ---- per-chain registers of the owner wavefront (uniform across its lanes)
  double f0 = 0.0, f1 = 0.0;
  long long abs_iter = 0, nacc = 0;
  int have_mean = 0, nerr = 0, status = FMCMC_CHAIN_OK;
  unsigned int bitword = 0;
  const bool mirror = (A.kind == FMCMC_KERNEL_NMIRROR || A.kind == FMCMC_KERNEL_UMIRROR);
  double obs_arate = fmh_nan();   // mirror kernels: lane = parameter (R's obs_arate turns into a k-vector through warm-up); th_prev: ans[i-2, ]
  double th_prev = 0.0;
  long long nzero = 0;            // rows 2..i-1 of this call equal to their predecessor (rowSums(diff(ans)^2) == 0)
  double* const Scur = L.SigA;   // ram: the factor S

  if (owner) {
    if (lane < k) {
      double t = A.theta0[cl * k + lane];
      L.th0[lane] = t;
      L.th1[lane] = t;
    }
    if (mirror) {
      if (lane < k) {
        L.mmu[lane] = A.fresh ? A.mu[lane] : A.mirror_mu[cl * k + lane];
        L.msc[lane] = A.fresh ? A.scale[lane] : A.mirror_scale[cl * k + lane];
      }
      if (!A.fresh) { abs_iter = A.abs_iter[cl]; obs_arate = A.obs_arate[cl * k + (lane < k ? lane : 0)]; }
    }
    if (adaptive) {
      if (A.fresh) {
        for (int e = lane; e < kf * LD; e += 64) {
          int a = e / LD, b = e % LD;
          L.SigA[e] = (a == b) ? 1.0 * A.eps : 0.0;
          if (A.kind == FMCMC_KERNEL_ADAPT) L.SigB[e] = 0.0;
        }
      } else {
        for (int e = lane; e < kf * LD; e += 64) {
          int a = e / LD, b = e % LD;
          // (ram: S is a LOWER factor; whatever the caller left above the diagonal is not part of it)
          L.SigA[e] = (b < kf && (b <= a || A.kind == FMCMC_KERNEL_ADAPT)) ? A.Sigma[(cl * kf + a) * kf + b] : 0.0;
          if (A.kind == FMCMC_KERNEL_ADAPT) L.SigB[e] = 0.0;
        }
        abs_iter = A.abs_iter[cl];
        if (A.nerrors) nerr = A.nerrors[cl];
        if (A.kind == FMCMC_KERNEL_ADAPT) {
          have_mean = A.have_mean[cl];
          if (lane < kf) L.vmp[lane] = A.mean_prev[cl * kf + lane];
        }
      }
    }
  }
  __syncthreads();

  // ---- row 1: f0 = f(initial)
  evaluate();
  __syncthreads();
  const long long S = A.S;
  // row bookkeeping without integer division: `thin_ctr` counts rows since the last kept one
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  int thin_ctr = 0;       // (r - burnin) mod thin for r > burnin
  long long srow = 0;     // next kept-row index
  double* const out_s = A.samples + (cl * k + (lane < k ? lane : 0)) * A.ldS;
  double* const out_d = A.draws ? A.draws + (cl * k + (lane < k ? lane : 0)) * A.ldS : nullptr;
  double* const out_l = A.logpost ? A.logpost + cl * A.ldS : nullptr;
  auto store_row = [&](int r, double lpv) {
    if (r > burnin) {
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (lane < k) {
          out_s[srow] = L.th0[lane];
          if (out_d) out_d[srow] = L.th1[lane];
        }
        if (out_l && lane == 0) out_l[srow] = lpv;
        srow += 1;
      }
    }
  };
  if (owner) {
    f0 = finish_logpost<FAM>(A, L.th1, total_of(myc), s_hs);
    f1 = f0;
    if (lane < kf) L.vrs[lane] = L.th0[s_which[lane]];
    if (A.hist_rows > 0 && lane < kf) A.hist[((long long)cl * A.hist_rows + (1 % A.hist_rows)) * kf + lane] = L.th0[s_which[lane]];
    store_row(1, f0);
  }

  // variate `a` (a == kz: the log accept uniform) of loop step ii for chain slot c, into slot t of the tile
  auto draw_variate = [&](int c, int t, int a, long long ii) {
    if (c < ncw && ii <= A.nsteps) {
      const long long clc = cg0 + c;
      const unsigned int cg = (unsigned int)(A.chain_base + clc);
      const unsigned int st = (unsigned int)(A.step_base + ii);
      double v;
      if (a == kz) {
        v = (fed) ? A.fed_logu[clc * A.nsteps + (ii - 1)] : fmh_log_accept_u(A.seed, st, cg);
        s_lu[c * TB + t] = v;
      } else {
        if (fed) v = A.fed_z[(clc * A.nsteps + (ii - 1)) * kz + a];
        else if (A.kind == FMCMC_KERNEL_RAM && A.ram_df > 0.0) v = fmh_student_t(A.seed, st, cg, (unsigned int)a, A.ram_df);
        else if (A.variate == 1) v = fmh_unif(A.seed, st, cg, (unsigned int)a);
        else v = fmh_normal(A.seed, st, cg, (unsigned int)a);
        s_zt[(c * TB + t) * kz + a] = v;
      }
    }
  };
  const bool ring = (CW <= NW - 4) && TB >= 4;

  // ---- main loop
  int tt = -1;         // position inside the RNG tile
  int ord = 0;         // ordered scheme: (i - 1) mod kf
#ifdef FMCMC_STAMP
  for (int q = 0; q < 16; q++) stamps.acc[q] = 0;
  stamps.prev = stamp_clk();
#endif
  for (int i = 2; i <= nsteps; i++) {
    tt = (tt + 1 == TB) ? 0 : tt + 1;
    ord = (ord + 1 == kf) ? 0 : ord + 1;
    bool ram_gate = false;
    // ================= RNG: the variates of a step sit in slot (step - 2) mod TB of the LDS tile =================
    // Tile mode: every TB steps all 512 threads draw the next TB steps.  Ring mode (workgroups with at least four waves
    // that own no chain): the first tile as above, then the NON-owner waves refill one slot per step while the owners are
    // busy with the scalar phases (after the evaluation of step i: the slot of step i - 1, for step i - 1 + TB) -- the
    // Student-t variates of a wide kernel_ram sweep cost 1.3 us per step when everybody stops for them.
    if (tt == 0 && (!ring || i == 2)) {
      __syncthreads();  // owners are done with the previous tile (and with s_part)
      const int per_c = TB * (kz + 1);
      for (int idx = tid; idx < CW * per_c; idx += NT) {
        const int c = idx / per_c, rem = idx - c * per_c;
        const int t = rem / (kz + 1), a = rem - t * (kz + 1);
        draw_variate(c, t, a, (long long)i + t);
      }
      __syncthreads();
    }
    const double* zt = s_zt + ((owner ? myc : 0) * TB + tt) * kz;
    FMH_STAMP(stp, 0);
    // ================= scalar phase A: proposal =================
    if (owner && status == FMCMC_CHAIN_OK) {
      if (A.kind == FMCMC_KERNEL_NORMAL || A.kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) {
        if (lane < k) L.th1[lane] = L.th0[lane];
        wave_sync_lds();
        const bool refl = (A.kind == FMCMC_KERNEL_NORMAL_REFLECTIVE);
        // plan_update_sequence (R/kernel.R:66-133): every scheme but "joint" updates ONE parameter per step
        const bool single = (A.scheme != FMCMC_SCHEME_JOINT);
        int col = 0;
        if (A.scheme == FMCMC_SCHEME_ORDERED) {
          col = s_which[ord];
        } else if (A.scheme == FMCMC_SCHEME_EXPLICIT) {
          col = A.scheme_seq[(i - 1) % A.scheme_len];
        } else if (A.scheme == FMCMC_SCHEME_RANDOM) {
          if (fed) {
            col = A.scheme_cols[cl * A.nsteps + (i - 1)];
          } else {
            // sample(which(!fixed), nsteps, TRUE)[i]; a single free parameter at position j makes R sample from 1:j
            const unsigned int npool = (kf == 1) ? (unsigned int)(s_which[0] + 1) : (unsigned int)kf;
            const unsigned int idx = fmh_scheme_index(A.seed, (unsigned int)i, cgid, npool);
            col = (kf == 1) ? (int)idx : s_which[idx];
            if (A.scheme_cols && lane == 0) A.scheme_cols[cl * A.nsteps + (i - 1)] = col;
          }
        }
        const int nupd = single ? 1 : kf;
        if (lane < nupd) {
          int j = single ? col : s_which[lane];
          double z = zt[lane];
          double t = L.th0[j] + (s_mu[j] + s_scale[j] * z);
          if (refl) t = reflect1(t, s_lb[j], s_ub[j]);
          L.th1[j] = t;
        }
      } else if (mirror) {
        // R/kernel_mirror.R:66-131 (nmirror), :203-262 (umirror); twin of the oracle's propose_mirror
        if (abs_iter >= 1 && abs_iter <= A.warmup && lane < k)   // mu <<- mean_recursive(ans[i-1, ], mu, abs_iter)
          L.mmu[lane] = (L.mmu[lane] * (double)abs_iter + L.th0[lane]) / ((double)abs_iter + 1);
        if (abs_iter == A.nadapt) {   // the one-off scale adaptation (the closure reads its argument `nadapt`)
          obs_arate = 1.0 - (double)nzero / (double)(i - 2);
          const double num = fmh_tan_0_halfpi(1.5707963267948966 * obs_arate);
          const double den = fmh_tan_0_halfpi(1.5707963267948966 * A.arate);
          if (lane < k) L.msc[lane] = L.msc[lane] * num / den;
        } else if (abs_iter > A.nadapt && abs_iter <= A.warmup) {
          // obs_arate <<- mean_recursive(as.double(ans[i-1, ] != ans[i-2, ]), obs_arate, abs_iter), element-wise (R/kernel_mirror.R:108-118);
          // the first proposal of a call has no ans[i-2, ]: numeric(0) in R, NaN here (twin of the oracle's propose_mirror)
          const double xt = (lane < k && L.th0[lane] != th_prev) ? 1.0 : 0.0;
          obs_arate = (i < 3) ? fmh_nan() : (obs_arate * (double)abs_iter + xt) / ((double)abs_iter + 1);
        }
        if (lane < k) L.th1[lane] = L.th0[lane];
        wave_sync_lds();
        const bool single = (A.scheme != FMCMC_SCHEME_JOINT);
        int col = 0;
        if (A.scheme == FMCMC_SCHEME_ORDERED) {
          col = s_which[ord];
        } else if (A.scheme == FMCMC_SCHEME_EXPLICIT) {
          col = A.scheme_seq[(i - 1) % A.scheme_len];
        } else if (A.scheme == FMCMC_SCHEME_RANDOM) {
          if (fed) {
            col = A.scheme_cols[cl * A.nsteps + (i - 1)];
          } else {
            const unsigned int npool = (kf == 1) ? (unsigned int)(s_which[0] + 1) : (unsigned int)kf;
            const unsigned int idx = fmh_scheme_index(A.seed, (unsigned int)i, cgid, npool);
            col = (kf == 1) ? (int)idx : s_which[idx];
            if (A.scheme_cols && lane == 0) A.scheme_cols[cl * A.nsteps + (i - 1)] = col;
          }
        }
        const int nupd = single ? 1 : kf;
        if (lane < nupd) {
          const int j = single ? col : s_which[lane];
          const double z = zt[lane];
          double t;
          if (A.kind == FMCMC_KERNEL_NMIRROR) {
            t = (2.0 * L.mmu[j] - L.th0[j]) + L.msc[j] * z;
          } else {   // runif(k, 2 mu - theta[which.] -+ sqrt3 scale): mu / scale of the a-th updated parameter are [a], as in R
            const double sqrt3 = fmh_sqrt(3.0);
            const double c = 2.0 * L.mmu[lane] - L.th0[j];
            const double lo = c - sqrt3 * L.msc[lane], hi = c + sqrt3 * L.msc[lane];
            t = lo + (hi - lo) * z;
          }
          L.th1[j] = reflect1(t, s_lb[j], s_ub[j]);
        }
        abs_iter += 1;
      } else if (A.kind == FMCMC_KERNEL_ADAPT) {
        // R/kernel_adapt.R:117-166
        if (A.until > (double)abs_iter && abs_iter > A.warmup && i > 2 && (i % A.freq) == 0) {
          const int H = A.hist_rows;
          double* ring = A.hist + (long long)cl * H * kf;
          // the ring rows were stored by other lanes of THIS wave: their stores acknowledged (s_waitcnt vmcnt(0)) is all it takes -- the
          // agent-scope fence that stood here wrote back and invalidated the XCD's caches at every adaptation (option audit:
          // kernel_adapt(freq = 2) at 14.7 us per step against 3.3 for freq = 1)
          if (H > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          if (A.bw > 0) {
            // windowed AM: Sigma <<- Sd * (cov(ans[(i - bw + 1):(i - 1), which.]) + Ik) :120-125 (twin of the oracle's canonical cov)
            const int N = A.bw - 1;
            if (i - A.bw + 1 < 1 || N < 2) {
              status = FMCMC_CHAIN_BAD_WINDOW;
            } else {
              double m = 0.0;
              if (lane < kf) {
                double sm = 0.0;
                for (int r = i - A.bw + 1; r <= i - 1; r++) sm = sm + ring[(long long)(r % H) * kf + lane];
                m = sm / (double)N;
                for (int b = 0; b < kf; b++) L.SigA[lane * LD + b] = 0.0;
              }
              for (int r = i - A.bw + 1; r <= i - 1; r++) {
                double d = 0.0;
                if (lane < kf) {
                  d = ring[(long long)(r % H) * kf + lane] - m;
                  L.vv[lane] = d;
                }
                wave_sync_lds();
                if (lane < kf)
                  for (int b = 0; b < kf; b++) L.SigA[lane * LD + b] = fmh_fma(d, L.vv[b], L.SigA[lane * LD + b]);
                wave_sync_lds();
              }
              if (lane < kf)
                for (int b = 0; b < kf; b++) {
                  const double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                  L.SigA[lane * LD + b] = A.Sd * (L.SigA[lane * LD + b] / (double)(N - 1) + ik);
                }
              wave_sync_lds();
            }
          } else if (i - A.freq < 1) {
            status = FMCMC_CHAIN_BAD_WINDOW;   // R: ans[0:(i-1), ] has fewer than freq rows, `[, , freq]` is out of bounds
          } else {
            // rows (i - freq):(i - 1) folded in one by one, t. = abs_iter - freq + (row - 1) (R/recursive.R:79-108,:129-136)
            for (int jr = 0; jr < A.freq; jr++) {
              const double t = (double)(abs_iter - A.freq + jr);
              double x = 0, mp = 0, mt = 0;
              if (lane < kf) {
                x = (A.freq == 1) ? L.th0[s_which[lane]] : ring[(long long)((i - A.freq + jr) % H) * kf + lane];
                mp = have_mean ? L.vmp[lane] : (L.vrs[lane] / (double)(i - 1));
                mt = (mp * t + x) / (t + 1);
                L.vv[lane] = x;
                L.vmp[lane] = mp;
                L.vmt[lane] = mt;
              }
              wave_sync_lds();
              if (lane < kf) {
                const double c1 = (t - 1) / t, c2 = 1.0 / t;
                for (int b = 0; b < kf; b++) {
                  double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                  double inner = t * (mp * L.vmp[b]) - (t + 1) * (mt * L.vmt[b]) + x * L.vv[b] + 1e-5 * ik;
                  L.SigA[lane * LD + b] = c1 * L.SigA[lane * LD + b] + c2 * inner;
                }
              }
              wave_sync_lds();
              if (lane < kf) L.vmp[lane] = mt;
              have_mean = 1;
            }
          }
        }
        abs_iter += 1;
        // root-free factor Sigma = L D L^T, left-looking, lane = row (twin of the oracle's ldl_lower_canon): L in the lower triangle of
        // SigB, the numerators W_ib = L_ib D_b in its upper one (W_ib at [b][i]), D in vv
        bool notpd = false;
        for (int j = 0; j < kf && status == FMCMC_CHAIN_OK; j++) {
          double s = 0.0;
          if (lane >= j && lane < kf) {
            s = L.SigA[lane * LD + j];
            for (int b = 0; b < j; b++) s = fmh_fma(-L.SigB[lane * LD + b], L.SigB[b * LD + j], s);
          }
          double d = shfl_d(s, j);
          if (!(d > 0.0) || !fmh_isfinite(d)) { notpd = true; break; }
          if (lane == j) { L.SigB[j * LD + j] = 1.0; L.vv[j] = d; }
          else if (lane > j && lane < kf) { L.SigB[lane * LD + j] = s / d; L.SigB[j * LD + lane] = s; }
          wave_sync_lds();
        }
        if (notpd) {
          status = FMCMC_CHAIN_NOT_PD;
        } else if (status == FMCMC_CHAIN_OK) {
          if (lane < k) L.th1[lane] = L.th0[lane];
          if (lane < kf) L.vz[lane] = fmh_sqrt(L.vv[lane]) * zt[lane];    // u = sqrt(D) z
          wave_sync_lds();
          if (lane < kf) {
            double s = 0.0;
            for (int b = 0; b <= lane; b++) s = fmh_fma(L.SigB[lane * LD + b], L.vz[b], s);
            int j = s_which[lane];
            double t = L.th0[j] + (s_mu[j] + s);
            L.th1[j] = reflect1(t, s_lb[j], s_ub[j]);
          }
        }
      } else {  // RAM, R/kernel_ram.R:123-126
        {
          const double s = ram_propose_rows((lds_dptr_t)Scur, (lds_dptr_t)L.SigB, (lds_dptr_t)zt, LD, kf);
          if (lane < kf) {
            const int j = s_which[lane];
            L.th1[j] = L.th0[j] + s;
          }
        }
        ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && (i % A.freq) == 0);
      }
      if (status != FMCMC_CHAIN_OK) {  // raised inside the proposal (NOT_PD)
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (lane < k) A.status_theta[cl * k + lane] = L.th1[lane];
      }
    }
    FMH_STAMP(stp, 1);
    __syncthreads();
    FMH_STAMP(stp, 2);
    // ================= collective evaluation of f(theta1) =================
    evaluate();
    FMH_STAMP(stp, 7);
    __syncthreads();
    FMH_STAMP(stp, 8);
    if (ring && wave >= CW && i >= 3) {   // (the owners last read the slot of step i - 1 before this barrier)
      const int t = (tt == 0) ? TB - 1 : tt - 1;
      for (int idx = tid - CW * 64; idx < CW * (kz + 1); idx += (NW - CW) * 64)
        draw_variate(idx / (kz + 1), t, idx % (kz + 1), (long long)i - 1 + TB);
    }
    // ================= scalar phase B: RAM adaptation (needs f(theta1) un-reflected) =================
    double f1_pre = 0.0;      // f(theta1) when phase B has it already (unbounded kernel_ram: the proposal is final)
    bool have_f1 = false;
    if (A.kind == FMCMC_KERNEL_RAM) {
      bool changed = false;
      if (owner && status == FMCMC_CHAIN_OK) {
        if (ram_gate) {
          double f1u = finish_logpost<FAM>(A, L.th1, total_of(myc), s_hs);
          f1_pre = f1u;
          have_f1 = !A.ram_bounded;
          double a_n = fmh_exp(f1u - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          double eta = (double)kf * fmh_exp(A.ram_neg_exp * fmh_log((double)i));
          if (eta > 1.0) eta = 1.0;
          FMH_STAMP(stp, 11);
          const double zl = (lane < kf) ? zt[lane] : 0.0;
          const double Pj1 = lane_scan_wave(zl * zl);              // sum_{b <= lane} z_b^2
          double Pj = __shfl_up(Pj1, 1, 64);
          Pj = (lane == 0) ? 0.0 : Pj;
          const double nrm2 = readlane_d(Pj1, kf - 1);
          double cp = (eta * (a_n - A.arate)) / nrm2;
          FMH_STAMP(stp, 12);
          if (cp != 0.0 && fmh_isfinite(cp)) {
            double dl, kl;
            const bool okl = ram_coef(cp, Pj, Pj1, zl, dl, kl);
            if (__any(lane < kf && !okl)) {
              nerr += 1;
            } else {
              FMH_STAMP(stp, 13);
              if (lane < kf) { L.vmp[lane] = dl; L.vmt[lane] = kl; }    // (vmt == vmp + kf: d_j | kappa_j)
              wave_sync_lds();
              ram_update_rows((lds_dptr_t)Scur, (lds_dptr_t)L.SigB, (lds_dptr_t)L.vmp, LD, kf);
              wave_sync_lds();
              FMH_STAMP(stp, 14);
            }
          }
          if (A.constr) {  // Sigma <<- constr[which., which.] * Sigma (R/kernel_ram.R:149-150)
            if (lane < kf)
              for (int b = 0; b < kf; b++) Scur[lane * LD + b] = A.constr[lane * kf + b] * Scur[lane * LD + b];
            wave_sync_lds();
          }
        }
        abs_iter += 1;
        if (A.ram_bounded) {
          if (lane < kf) {
            int j = s_which[lane];
            double t0 = L.th1[j];
            double t1 = reflect1(t0, s_lb[j], s_ub[j]);
            if (!(t1 == t0)) { L.th1[j] = t1; changed = true; }
          }
          if (__any(changed)) s_flag[0] = 1;
        }
      }
      if (A.ram_bounded) {  // uniform over the workgroup (launch-time constant)
        __syncthreads();
        const bool again = (s_flag[0] != 0);
        __syncthreads();
        if (again) {
          if (tid == 0) s_flag[0] = 0;
          evaluate();
          __syncthreads();
        }
      }
    }
    FMH_STAMP(stp, 9);
    // ================= scalar phase C: accept / store (R/mcmc.R:754-778) =================
    if (owner && status == FMCMC_CHAIN_OK) {
      f1 = have_f1 ? f1_pre : finish_logpost<FAM>(A, L.th1, total_of(myc), s_hs);
      FMH_STAMP(stp, 15);
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (lane < k) A.status_theta[cl * k + lane] = L.th1[lane];
      } else {
        const double lu = s_lu[myc * TB + tt];
        bool moved = false;
        if (mirror && lane < k) th_prev = L.th0[lane];     // (row i - 1, the row before the one decided now)
        if (lu < ratio) {
          if (mirror) {   // rowSums(diff(ans)^2) of the row about to be stored (sequential sum, as in the oracle)
            double sq = 0.0;
            for (int a = 0; a < k; a++) sq = sq + (L.th1[a] - L.th0[a]) * (L.th1[a] - L.th0[a]);
            moved = (sq != 0.0);
            wave_sync_lds();
          }
          if (lane < k) L.th0[lane] = L.th1[lane];
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        if (mirror && !moved) nzero += 1;
        wave_sync_lds();
        store_row(i, f1);
        if (A.kind == FMCMC_KERNEL_ADAPT && lane < kf) L.vrs[lane] = L.vrs[lane] + L.th0[s_which[lane]];
        if (A.hist_rows > 0 && lane < kf)   // row i of ans[, which.] for the windowed / strided adaptation
          A.hist[((long long)cl * A.hist_rows + (i % A.hist_rows)) * kf + lane] = L.th0[s_which[lane]];
      }
    }
    if (owner && A.accept_bits && lane == 0 && (((i - 1) & 31) == 31 || i == nsteps)) {
      A.accept_bits[cl * (long long)((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
      bitword = 0;
    }
    FMH_STAMP(stp, 10);
  }
#ifdef FMCMC_STAMP
  if (wave == 0 && owner && lane < 16 && k >= 16) A.status_theta[cl * k + lane] = (double)stamps.acc[lane];
  // (narrow models: over the first 16 kept rows of the chain's first parameter -- the stamped build's results are not results)
  if (FMCMC_STAMP_WAVE >= 0 && wave == 0 && owner && lane < 16 && k < 16 && A.ldS >= 16) { __builtin_amdgcn_s_waitcnt(0); A.samples[(cl * k) * A.ldS + lane] = (double)stamps.acc[lane]; }
  if (FMCMC_STAMP_WAVE < 0 && lane < 16 && A.ldS >= 128) {   // every wave: rows 16 w .. 16 w + 15 of the workgroup's first chain
    __syncthreads();
    A.samples[(cg0 * k) * A.ldS + 16 * wave + lane] = (double)stamps.acc[lane];
  }
#endif

  // ---- write state back
  if (owner) {
    if (lane < k) A.theta0[cl * k + lane] = L.th0[lane];
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      // (a grid-wide hand-over of the observation-sharded evaluation was lost -- bit 31 of the epoch, eval_sharded --: EVERY chain
      //  of the workgroup says so.  Until round 4 the OK of the line below overwrote the one status word eval_partials had set:
      //  found by the forced fault of test_a_lost_hand_over_ends_in_status_5_not_in_a_hang)
      if (sh_epoch & 0x80000000u) { A.status[cl] = FMCMC_CHAIN_SYNC_TIMEOUT; A.status_step[cl] = 0; }
      else if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
      if (mirror) A.abs_iter[cl] = abs_iter;
      if (adaptive) {
        A.abs_iter[cl] = abs_iter;
        if (A.nerrors) A.nerrors[cl] = nerr;
        if (A.kind == FMCMC_KERNEL_ADAPT) A.have_mean[cl] = have_mean;
      }
    }
    if (mirror && lane < k) {
      A.mirror_mu[cl * k + lane] = L.mmu[lane];
      A.mirror_scale[cl * k + lane] = L.msc[lane];
      A.obs_arate[cl * k + lane] = obs_arate;
    }
    if (adaptive) {
      wave_sync_lds();
      const double* Sfin = (A.kind == FMCMC_KERNEL_RAM) ? Scur : L.SigA;
      for (int e = lane; e < kf * kf; e += 64) {
        int a = e / kf, b = e % kf;
        A.Sigma[(cl * kf + a) * kf + b] = Sfin[a * LD + b];
      }
      if (A.kind == FMCMC_KERNEL_ADAPT && lane < kf) A.mean_prev[cl * kf + lane] = L.vmp[lane];
    }
  }
}

}  // namespace
