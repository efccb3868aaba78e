// mh_mfma_ad.hpp -- mh_sweep_mfma_ad<KIND, NG, KX>: kernel_adapt / kernel_ram on the fp64-MFMA evaluation for data sets beyond
// the wave-specialised kernel's registers (round 4: n > 10240 at p <= 3, n > 5120 / 4096 at p <= 7 used to fall to the general
// kernel, 5-6x slower per flop -- profiles/r04_shape_map.md).
#pragma once

namespace {

// The evaluation is mh_sweep_mfma's EXT form (mh_mfma.hpp: v_mfma_f64_4x4x4 with A = [x_1 .. x_p, y], B = [b_1 .. b_p, -1],
// C = b_0 for 16 observations x 4 chains, the canonical tree's levels 1..32 inside the wave): NSR observation slots resident
// in operand registers, the rest streamed every step from the operand-order copy (mfma_build_stream).  All four chains of the
// workgroup are evaluated together by all eight waves; then waves 0..3 run ONE step each of the register-row adaptive owner
// of mh_sweep_spec (spec_owner_adaptive_reg: recursive covariance + Cholesky, or the product-form RAM update, rows in VGPRs)
// between two workgroup barriers.  At these sizes the evaluation is most of a step, so the exposed owner phase costs what it
// costs the normal kernels on mh_sweep_mfma; below them mh_sweep_spec, which overlaps owners and evaluation, stays the kernel.
template <int NG>
struct MfmaAdShape {
  static constexpr int NSR = (NG == 1) ? 12 : (NG == 2 ? 6 : 3);    // resident slots (operand registers: 4 NSR NG doubles per lane)
  static constexpr int RD = (NG == 1) ? 4 : 2;      // streamed slots in flight per wave
};

// the owners' side of a step (spec_owner_adaptive_reg's SYNC policy): they evaluate like every wave, then meet on barriers
template <class EV>
struct MfmaAdSync {
  static constexpr bool PREP_EARLY = true;
  EV& ev; const double* s_fold; int myc; unsigned* s_need;
  __device__ __forceinline__ double fold() const {
    double wsum = s_fold[myc * NW + (threadIdx.x & 7)];
    wsum = wsum + dpp_d<0xB1>(wsum);            // canonical levels 64, 128, 256 over the eight wave sums
    wsum = wsum + dpp_d<0x4E>(wsum);
    return wsum + dpp_d<0x141>(wsum);
  }
  // bounded kernel_ram (spec_owner_adaptive_reg<.., BND>): one more barrier per step tells the workgroup whether any of its
  // chains' proposals was moved by the reflection; if so all eight waves evaluate the four (republished) proposals once more
  template <class W> __device__ __forceinline__ bool second(bool need, W&& republish, double& tot2) const {
    if (need && (threadIdx.x & 63) == 0) lds_st_u32(s_need, 1u);
    lds_barrier();
    const bool again = __builtin_amdgcn_readfirstlane((int)lds_ld_u32(s_need)) != 0;
    if (!again) return false;
    republish();
    lds_barrier();                              // (every wave has read the flag; the proposals are in place)
    if (threadIdx.x == 0) lds_st_u32(s_need, 0u);
    ev();
    lds_barrier();
    tot2 = fold();
    return true;
  }
  __device__ __forceinline__ void second_idle() const {
    double t;
    (void)second(false, []() {}, t);
  }
  template <class F> __device__ __forceinline__ double total(int, F&& prep) const {
    ev();
    prep();                                     // the sigma-only half of the closed form, in the slack before the barrier
    lds_barrier();
    return fold();
  }
  __device__ __forceinline__ void publish(int) const { lds_barrier(); }
  __device__ __forceinline__ void final() const { lds_barrier(); }
};

// The mirror kernels' owner (kernel_nmirror / kernel_umirror, R/kernel_mirror.R:66-131, :203-262; twin of the general kernel's
// mirror branch, mh_streamed.hpp, and of the oracle's propose_mirror): joint scheme, no fixed parameter, lane = parameter with
// theta, the running mean mu and the scale in registers.  Until round 4 these kernels ran on the all-family kernel only
// (tools/option_audit.py: 17.8 us per step at C2's shape).
template <int KIND, class SYNC>
__device__ __forceinline__ void mfma_owner_mirror(const SweepArgs& A, int myc, int cl, double* s_th1, SYNC& sync) {
  const int lane = threadIdx.x & 63;
  const int k = A.k, kz = A.kz;
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const bool rl = lane < k;
  const int jl = rl ? lane : 0;
  const double lb_l = A.lb[jl], ub_l = A.ub[jl];
  double th0 = rl ? A.theta0[(long long)cl * k + lane] : 0.0, th1 = th0;
  double mmu = rl ? (A.fresh ? A.mu[jl] : A.mirror_mu[(long long)cl * k + jl]) : 0.0;
  double msc = rl ? (A.fresh ? A.scale[jl] : A.mirror_scale[(long long)cl * k + jl]) : 0.0;
  double f0 = 0.0, obs_arate = fmh_nan(), th_prev = 0.0;   // (obs_arate: lane = parameter -- R's turns into a k-vector through warm-up; th_prev: ans[i-2, ])
  long long abs_iter = 0, nzero = 0;
  if (!A.fresh) { abs_iter = A.abs_iter[cl]; obs_arate = A.obs_arate[(long long)cl * k + jl]; }
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0;
  unsigned int bitword = 0;
  char* const s_base = reinterpret_cast<char*>(A.samples) + ((long long)cl * k) * A.ldS * 8;
  char* const d_base = A.draws ? reinterpret_cast<char*>(A.draws) + ((long long)cl * k) * A.ldS * 8 : nullptr;
  char* const l_base = A.logpost ? reinterpret_cast<char*>(A.logpost) + (long long)cl * A.ldS * 8 : nullptr;
  const unsigned int lane_off = (unsigned int)((long long)jl * A.ldS * 8);
  unsigned int srow8 = 0;
  const char* const z_base = reinterpret_cast<const char*>(A.fed_z) + ((long long)cl * nsteps) * kz * 8;
  const unsigned int z_lane = (unsigned int)(jl) * 8u;
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double { return *reinterpret_cast<const double*>(z_base + (z_lane + (unsigned int)row * (unsigned int)(kz * 8))); };
  double z_nx = (rl && nsteps >= 2) ? ld_z(1) : 0.0;
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };
  auto logpost_of = [&](double tot, double sigma) -> double {   // Gaussian linreg closed form (as the other owners')
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) f = fmh_nan();
    else if (sigma == 0.0) f = -fmh_inf();
    else {
      double t1 = fmh_log(sigma) + FMH_K(FMH_LN_SQRT_2PI);
      double q = (0.5 * tot) / (sigma * sigma);
      f = -(dn * t1) - q;
    }
    if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
    return f;
  };
  for (int v = 1; v <= nsteps; v++) {
    const double tot = sync.total(v, []() {});
    const double f1 = logpost_of(tot, readlane_d(th1, k - 1));
    bool st_row = false;
    double st_th0 = 0.0;
    const double st_dr = th1;
    if (v == 1) {
      f0 = f1;
      if (1 > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
    } else if (status == FMCMC_CHAIN_OK) {
      const int i = v;
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (rl) A.status_theta[(long long)cl * k + lane] = th1;
        flush_bits(i);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
        bool moved = false;
        th_prev = th0;                          // (row i - 1, the row before the one decided now)
        if (lu < ratio) {
          const double d = th1 - th0;           // rowSums(diff(ans)^2) of the row about to be stored: the sequential sum of the oracle
          double sq = 0.0;
          for (int a = 0; a < k; a++) { const double da = readlane_d(d, a); sq = sq + da * da; }
          moved = (sq != 0.0);
          th0 = th1;
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        if (!moved) nzero += 1;
        if (i > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
        if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
      }
    }
    // ---- proposal of loop step i = v + 1
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK) {
        const int i = v + 1;
        const double z = z_nx;
        z_nx = rl ? ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1) : 0.0;
        if (abs_iter >= 1 && abs_iter <= A.warmup) mmu = (mmu * (double)abs_iter + th0) / ((double)abs_iter + 1);   // mean_recursive(ans[i-1, ], mu, abs_iter)
        if (abs_iter == A.nadapt) {   // the one-off scale adaptation (the closure reads its argument `nadapt`)
          obs_arate = 1.0 - (double)nzero / (double)(i - 2);
          const double num = fmh_tan_0_halfpi(1.5707963267948966 * obs_arate);
          const double den = fmh_tan_0_halfpi(1.5707963267948966 * A.arate);
          msc = msc * num / den;
        } else if (abs_iter > A.nadapt && abs_iter <= A.warmup) {
          // obs_arate <<- mean_recursive(as.double(ans[i-1, ] != ans[i-2, ]), obs_arate, abs_iter), element-wise (R/kernel_mirror.R:108-118,
          // :246-253); the first proposal of a call has no ans[i-2, ]: numeric(0) in R, NaN here (twin of the oracle's propose_mirror)
          obs_arate = (i < 3) ? fmh_nan() : (obs_arate * (double)abs_iter + ((th0 != th_prev) ? 1.0 : 0.0)) / ((double)abs_iter + 1);
        }
        double t;
        if (KIND == FMCMC_KERNEL_NMIRROR) {
          t = (2.0 * mmu - th0) + msc * z;
        } else {   // runif(k, 2 mu - theta -+ sqrt3 scale)
          const double sqrt3 = fmh_sqrt(3.0);
          const double c = 2.0 * mmu - th0;
          const double lo = c - sqrt3 * msc, hi = c + sqrt3 * msc;
          t = lo + (hi - lo) * z;
        }
        th1 = reflect1(t, lb_l, ub_l);
        abs_iter += 1;
        if (rl) s_th1[myc * PIPE_KMAX + lane] = th1;
      }
      sync.publish(v + 1);
    } else {
      sync.final();
    }
    if (st_row) {
      if (rl) {
        *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = st_th0;
        if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = st_dr;
      }
      if (l_base && lane == 0) *reinterpret_cast<double*>(l_base + srow8) = f1;
      srow8 += 8;
    }
  }
  if (rl) {
    A.theta0[(long long)cl * k + lane] = th0;
    A.mirror_mu[(long long)cl * k + lane] = mmu;
    A.mirror_scale[(long long)cl * k + lane] = msc;
    A.obs_arate[(long long)cl * k + lane] = obs_arate;
  }
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    A.abs_iter[cl] = abs_iter;
  }
}

// NSV: resident slots -- MfmaAdShape<NG>::NSR, or 1 for short data (512 < n <= 512 NSR: everything else streamed)
template <int KIND, int NG, int KX, bool BND = false, int NSV = MfmaAdShape<NG>::NSR>
__global__ __launch_bounds__(NT) void mh_sweep_mfma_ad(const SweepArgs A) {
  constexpr int CW = 4, NS = NSV, RD = MfmaAdShape<NG>::RD, TN = NS * 4, MB = TN < 12 ? TN : 12;
  static_assert(TN % MB == 0, "batches of MB pairs");
  static_assert(KX >= 0 || !BND, "the bounded kernel_ram has the register-row owner only");   // (KX = -1: matrices in LDS, -2: the mirror kernels' owner)
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX]
  double* s_fold = s_th1 + CW * PIPE_KMAX;         // [CW][NW] per-wave sums of every chain (canonical levels 1..32 done)
  unsigned* s_need = reinterpret_cast<unsigned*>(s_fold + CW * NW);   // BND: "a reflection moved a proposal of this workgroup in this step"
  double* s_par = s_fold + CW * NW + 1;            // KX < 0: [4][PIPE_KMAX] mu | scale | lb | ub, then [CW][SPEC_ADS] the owners' matrices
  double* s_ad = s_par + 4 * PIPE_KMAX;
  if (tid == 0) lds_st_u32(s_need, 0u);
  if (KX == -1 && tid < PIPE_KMAX) {
    const bool in = tid < A.k;
    s_par[0 * PIPE_KMAX + tid] = in ? A.mu[tid] : 0.0;
    s_par[1 * PIPE_KMAX + tid] = in ? A.scale[tid] : 0.0;
    s_par[2 * PIPE_KMAX + tid] = in ? A.lb[tid] : 0.0;
    s_par[3 * PIPE_KMAX + tid] = in ? A.ub[tid] : 0.0;
  }
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, ic = A.intercept, P = A.p, next = A.mf_next;

  // ---- A operands of the resident slots (see mh_sweep_mfma)
  const int feat = lane >> 4, o16 = lane & 15;
  const int cl_a = 16 * (o16 & 3) + 4 * (o16 >> 2);
  double areg[NG][TN];
#pragma unroll
  for (int q = 0; q < NG; q++) {
    const int f = 4 * q + feat;
#pragma unroll
    for (int t = 0; t < TN; t++) {
      const int sl = t >> 2, g = t & 3;
      const long long i = (long long)(64 * wave + cl_a + g) + (long long)NT * sl;
      double a = 0.0;
      if (i < A.n) {
        if (f < P) a = A.X[(long long)f * A.n + i];
        else if (f == P) a = A.y[i];
      }
      areg[q][t] = a;
    }
  }
  const int jch = lane & 3;
  const int cl_d = 16 * (lane >> 4) + 4 * ((lane >> 2) & 3);
  unsigned vbits = 0;    // validity of this lane's 4 results in the LAST (streamed) slot; every earlier slot is full
#pragma unroll
  for (int g = 0; g < 4; g++)
    if ((long long)(64 * wave + cl_d + g) + (long long)NT * (NS - 1 + next) < A.n) vbits |= 1u << g;
  if (tid < CW * PIPE_KMAX) {
    const int c = tid / PIPE_KMAX, j = tid - c * PIPE_KMAX;
    s_th1[tid] = (c < ncw && j < k) ? A.theta0[(cg0 + c) * k + j] : 0.0;
  }
  lds_barrier();

  typedef double mf_d4 __attribute__((ext_vector_type(4)));
  const mf_d4* sp = reinterpret_cast<const mf_d4*>(A.mf_stream) + ((long long)wave * next * NG) * 64 + lane;
  // evaluation of the published proposals of all 4 chains; leaves this wave's sums in s_fold
  auto evaluate = [&]() {
    const double* tj = s_th1 + jch * PIPE_KMAX;
    double bop[NG];
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int f = 4 * q + feat;
      bop[q] = (f < P) ? tj[ic + f] : (f == P ? -1.0 : 0.0);
    }
    const double cop = ic ? tj[0] : 0.0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    mf_d4 ring[RD][NG];                          // the first streamed slots are requested before the resident ones run
#pragma unroll
    for (int r = 0; r < RD; r++)
#pragma unroll
      for (int q = 0; q < NG; q++) ring[r][q] = sp[(((r < next) ? r : next - 1) * NG + q) * 64];
#pragma unroll
    for (int t0 = 0; t0 < TN; t0 += MB) {
      double d[MB];
#pragma unroll
      for (int u = 0; u < MB; u++) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[0][t0 + u], bop[0], cop, 0, 0, 0);
#pragma unroll
      for (int q = 1; q < NG; q++)
#pragma unroll
        for (int u = 0; u < MB; u++) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[q][t0 + u], bop[q], d[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < MB; u++) acc[u & 3] = fmh_fma(d[u], d[u], acc[u & 3]);
    }
    double cml[4];                               // C operands of the LAST slot: 0 where it is padding (-r == 0 exactly)
#pragma unroll
    for (int g = 0; g < 4; g++) cml[g] = ((vbits >> g) & 1u) ? cop : 0.0;
    for (int e0 = 0; e0 < next; e0 += RD) {
#pragma unroll
      for (int r = 0; r < RD; r++) {
        const int e = e0 + r;
        if (e < next) {                          // (uniform)
          mf_d4 a[NG];
#pragma unroll
          for (int q = 0; q < NG; q++) a[q] = ring[r][q];
          const int en = (e + RD < next) ? e + RD : next - 1;
#pragma unroll
          for (int q = 0; q < NG; q++) ring[r][q] = sp[(en * NG + q) * 64];
          const bool last = (e == next - 1);
          double d[4];
#pragma unroll
          for (int g = 0; g < 4; g++) d[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0][g], bop[0], last ? cml[g] : cop, 0, 0, 0);
#pragma unroll
          for (int q = 1; q < NG; q++)
#pragma unroll
            for (int g = 0; g < 4; g++) d[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[q][g], bop[q], d[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; g++) acc[g] = fmh_fma(d[g], d[g], acc[g]);
        }
      }
    }
    double fs = (acc[0] + acc[1]) + (acc[2] + acc[3]);        // canonical levels 1, 2 ... 32 (mh_sweep_mfma)
    fs = fs + dpp_d<0x114>(fs);
    fs = fs + dpp_d<0x118>(fs);
    {
      const unsigned long long u = (unsigned long long)__double_as_longlong(fs);
      const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
      const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      fs = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
           __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    }
    {
      const unsigned long long u = (unsigned long long)__double_as_longlong(fs);
      const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
      const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      fs = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
           __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    }
    if (lane >= 60) s_fold[jch * NW + wave] = fs;
  };

  const int myc = wave;
  MfmaAdSync<decltype(evaluate)> sync{evaluate, s_fold, myc < ncw ? myc : 0, s_need};
  if (myc >= ncw) {                               // waves without a chain: evaluate, meet the barriers of every step
    for (int v = 1; v <= nsteps; v++) {
      evaluate();
      lds_barrier();
      if constexpr (BND) sync.second_idle();
      lds_barrier();
    }
    return;
  }
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + myc);
  // KX < 0: the owners with their matrices in LDS (spec_owner_adaptive: any k <= 16, fixed parameters) -- 8 .. 15 covariates, or a
  // fixed parameter, beyond the wave-specialised kernel's range
  // (measured and dropped at the end of round 4: the register-row owner with rows of 16 -- level at 8 .. 11 covariates, 15.0 against
  //  15.3 us per step at n = 1e4, and 1.3 - 1.7x SLOWER at 12 .. 15, where four operand groups leave the rows no registers)
  if constexpr (KIND == FMCMC_KERNEL_NMIRROR || KIND == FMCMC_KERNEL_UMIRROR) mfma_owner_mirror<KIND>(A, myc, cl, s_th1, sync);
  else if constexpr (KX < 0) spec_owner_adaptive<KIND>(A, myc, cl, s_th1, s_par, sync, s_ad + myc * SPEC_ADS);
  else spec_owner_adaptive_reg<KIND, KX, decltype(sync), BND>(A, myc, cl, s_th1, sync);
}

size_t mfma_ad_lds_bytes(bool lds_owner) { return sizeof(double) * ((size_t)4 * PIPE_KMAX + 4 * NW + 1 + (lds_owner ? 4 * PIPE_KMAX + 4 * SPEC_ADS : 0)); }

}  // namespace
