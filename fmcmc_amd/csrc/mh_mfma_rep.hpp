// mh_mfma_rep.hpp -- mh_sweep_mfmar<KIND, NG, NS, DBG>: the fp64-MFMA kernel with the chain state REPLICATED in every
// wavefront (no owner waves, one workgroup barrier per step).
#pragma once

namespace {

// ==============================================================================================
// Same evaluation as mh_sweep_mfma (mh_mfma.hpp: operand layout, canonical lanes, padding through the C operand), but
// the scalar part of a step is no longer done by one owner wave per chain behind two barriers and two LDS round trips
// (lane partials out, proposals back).  In mh_sweep_mfma that owner phase -- fold of 512 lane partials, closed form,
// decision, proposal -- was ~1800 of the ~5300 ticks of a step with the MFMA pipe idle, and it cannot be overlapped:
// all four chains of a workgroup share the same MFMAs, and 1024 chains on 256 CUs leave no second group per CU.
//
// Here every lane of every wave carries the state of "its" chain j = lane % 4 (the chain whose B operand it feeds and
// whose results it receives): theta0/theta1 of the parameters it needs (B operands, intercept = C operand, sigma), f0,
// the variates of the next step.  A step is
//     MFMAs -> in-wave fold of the wave's 64 canonical lanes (levels 1..32 of the canonical tree, on
//              v_permlane16/32_swap + DPP, all four chains at once) -> 1 double per chain and wave to LDS
//     barrier (the only one)
//     every lane: 8 wave partials of its chain -> levels 64..256 -> closed form -> decision -> proposal, all in
//              registers, bit-identical in every lane of a chain -> the next B / C operands without touching LDS.
// The decisions are replicated, not communicated: same inputs, same instructions, same bits.  Waves 0..3 additionally
// (a) evaluate the sigma-only half of the closed form (log sigma, sigma^2) of the NEXT barrier in their MFMA slack
// and publish it through LDS, (b) store the rows of chain `wave`.
//
// Canonical tree inside a wave: result lane L holds chain j = L & 3 of canonical lane 64 w + 16 g + 4 a + b with
// b = L >> 4, a = (L >> 2) & 3, g = accumulator index.  Levels 1, 2 pair b (lanes L ^ 16, L ^ 32), levels 4, 8 pair a
// (L ^ 4, L ^ 8), levels 16, 32 pair g, levels 64..256 pair the waves.  Levels 1 and 2 run as a reduce-scatter over
// the four accumulators (a swap moves two registers' halves at once), which leaves accumulator g in row g of ONE
// register; levels 16 and 32 are then again L ^ 16 and L ^ 32.
// ==============================================================================================

template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_bank_d(double old, double v) {   // lanes of the banks in BANK take v[CTRL], others keep old
  unsigned long long u = (unsigned long long)__double_as_longlong(v), o = (unsigned long long)__double_as_longlong(old);
  unsigned lo = __builtin_amdgcn_update_dpp((unsigned)o, (unsigned)u, CTRL, 0xf, BANK, false);
  unsigned hi = __builtin_amdgcn_update_dpp((unsigned)(o >> 32), (unsigned)(u >> 32), CTRL, 0xf, BANK, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// x' + y' after v_permlane16_swap(x, y): rows (16 lanes) 0..3 of the sum = x0+x1, y0+y1, x2+x3, y2+y3
__device__ __forceinline__ double swap16_add(double x, double y) {
  unsigned long long ux = (unsigned long long)__double_as_longlong(x), uy = (unsigned long long)__double_as_longlong(y);
  auto rl = __builtin_amdgcn_permlane16_swap((unsigned)ux, (unsigned)uy, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
  double a = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
  double b = __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
  return a + b;
}
// x' + y' after v_permlane32_swap(x, y): halves (32 lanes) of the sum = x.lo + x.hi, y.lo + y.hi
__device__ __forceinline__ double swap32_add(double x, double y) {
  unsigned long long ux = (unsigned long long)__double_as_longlong(x), uy = (unsigned long long)__double_as_longlong(y);
  auto rl = __builtin_amdgcn_permlane32_swap((unsigned)ux, (unsigned)uy, false, false);
  auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
  double a = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
  double b = __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
  return a + b;
}
// levels 1..32 of the canonical tree over the wave's 64 canonical lanes, for the chain of every lane
__device__ __forceinline__ double mfma_wave_fold(double a0, double a1, double a2, double a3) {
  const double t0 = swap16_add(a0, a1);      // level 1 (b ^ 1): rows = g0, g1, g0, g1
  const double t1 = swap16_add(a2, a3);      //                  rows = g2, g3, g2, g3
  double u = swap32_add(t0, t1);             // level 2 (b ^ 2): row r = accumulator g = r, summed over b
  {                                          // level 4 (a ^ 1): lane ^ 4
    double o = dpp_bank_d<0x104, 0x5>(u, u);   // row_shl:4 -> banks 0, 2 read lane + 4
    o = dpp_bank_d<0x114, 0xA>(o, u);          // row_shr:4 -> banks 1, 3 read lane - 4
    u = u + o;
  }
  u = u + dpp_d<0x128>(u);                   // level 8 (a ^ 2): row_ror:8 == lane ^ 8
  u = swap16_add(u, u);                      // level 16 (g ^ 1)
  u = swap32_add(u, u);                      // level 32 (g ^ 2)
  return u;
}

// Timing ablations for tools/exp_mfmar.hip (never set in the product build; they change the results):
//   1 no row stores   2 no sigma-only publisher   4 no in-wave fold   8 no cross-wave butterfly   16 no general path
//   32 no MFMAs       64 no decision (always reject)
#ifndef MFR_X
#define MFR_X 0
#endif

constexpr int MFR_WP = 2 * 4 * 8;     // [parity][chain][wave] wave partials
constexpr int MFR_CF = 2 * 4 * 6;     // [parity][chain]{n (log sigma + ln sqrt 2 pi), sigma^2 (0: general path), ~0.5 / sigma^2, ~1 / sigma^2, sigma, -}
constexpr int MFR_RNG = 2 * 64;       // [parity][chain][1 + kz]: log-uniform and variates of the step, staged by wave 7

template <int KIND, int NG, int NS, bool DBG>
__global__ __launch_bounds__(NT) void mh_sweep_mfmar(const SweepArgs A) {
  constexpr int CW = 4;
  constexpr int MB = 8;             // (slot, lane group) pairs per batch = 2 observation slots
  constexpr int TN = NS * 4;        // pairs held per lane and group (NG * TN <= MF_NMF registers)
  constexpr int HS = NG + 2;        // parameter slots a lane carries: NG B operands, the intercept (C operand), sigma
  static_assert(NG * TN <= MF_NMF, "operand registers");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz;
  double* s_wp = smem;                 // [2][4][8]
  double* s_cf = s_wp + MFR_WP;        // [2][4][6]
  double* s_rng = s_cf + MFR_CF;       // [2][64]
  double* s_bnd = s_rng + MFR_RNG;     // [2][PIPE_KMAX] lb, ub (reflective kernels)
  if (tid < k) { s_bnd[tid] = A.lb[tid]; s_bnd[PIPE_KMAX + tid] = A.ub[tid]; }
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int ic = A.intercept;
  const int P = A.p;

  // ---- A operand: feature lane / 16 of observation (64 w + 16 g + lane % 16) + 512 s, for t = 4 s + g
  const int feat = lane >> 4, o16 = lane & 15;
  double areg[NG][TN];
#pragma unroll
  for (int q = 0; q < NG; q++) {
    const int f = 4 * q + feat;
#pragma unroll
    for (int t = 0; t < TN; t++) {
      const int sl = t >> 2, g = t & 3;
      const long long i = (long long)(64 * wave + 16 * g + o16) + (long long)NT * sl;
      double a = 0.0;
      if (i < A.n) {
        if (f < P) a = A.X[(long long)f * A.n + i];
        else if (f == P) a = A.y[i];
      }
      areg[q][t] = a;
    }
  }
  const int jch = lane & 3;
  const int cl_in_g = 4 * ((lane >> 2) & 3) + (lane >> 4);
  unsigned vbits = 0;    // validity of this lane's 4 results in the LAST slot (all earlier slots are full)
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int l = 64 * wave + 16 * g + cl_in_g;
    if ((long long)l + (long long)NT * (NS - 1) < A.n) vbits |= 1u << g;
  }

  // ---- the lane's chain and its parameter slots
  const bool cvalid = jch < ncw;
  const int cl = (int)cg0 + (cvalid ? jch : 0);           // local chain index (clamped: every address stays valid)
  const bool duty = cvalid && (wave == jch);              // this wave stores the rows of chain `wave`
  const int arow = (lane >> 2) & 3;
  const bool lead = duty && feat == 0 && arow == 3;       // logpost, accept bits, status, final scalars
  double th0[HS], th1[HS];
  bool sten[HS];
  unsigned int sdoff[HS];
  int pix[HS], zix[HS];
#pragma unroll
  for (int s = 0; s < HS; s++) {
    int pi = -1;
    double cst = 0.0;
    bool st = false;
    if (s < NG) {
      const int f = 4 * s + feat;
      if (f < P) pi = ic + f;
      else if (f == P) cst = -1.0;                        // the y column
      st = duty && arow == 0;
    } else if (s == NG) {
      if (ic) pi = 0;
      st = duty && feat == 0 && arow == 1;
    } else {
      pi = k - 1;
      st = duty && feat == 0 && arow == 2;
    }
    pix[s] = pi;
    const int pj = pi < 0 ? 0 : pi;
    const bool fx = (pi < 0) || A.fixed[pj] != 0;
    sten[s] = st && pi >= 0;
    int zi = 0;
    for (int j = 0; j < pj; j++) zi += A.fixed[j] ? 0 : 1;
    // a slot that is not updated adds entry 63 of the tile, a constant -0.0: x + (-0.0) == x bit for bit for every x
    zix[s] = fx ? 63 : jch * (kz + 1) + 1 + zi;           // the slot's increment in the staged tile
    sdoff[s] = (unsigned int)((((long long)cl * k + pj) * A.ldS) * 8);
    th0[s] = (pi >= 0) ? A.theta0[(long long)cl * k + pj] : cst;
    if (!cvalid && pi >= 0) th0[s] = 0.0;
    th1[s] = th0[s];
  }
  double f0 = 0.0;
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0;
  unsigned int srow8 = 0, bitword = 0;
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.ldS) * 8);
  const double dn = (double)A.n;
  // ---- random stream: ONE wave fetches the log-uniform and the kz variates of all four chains, one step ahead, with a
  // single load (lane = chain * (kz + 1) + item), turns the variates into the increments mu + scale z of "their"
  // parameter and stages everything in LDS in front of the barrier.  (Every lane loading its own copies -- 4 loads x
  // 8 waves per step, four cache lines per quad -- made the texture addresser the bottleneck of the step: 29 ms per
  // sweep instead of 22.)  The loader is one of the waves 0..3, which finish their MFMAs first.
  const bool loader = (wave == 3);
  const int l_c = lane / (kz + 1), l_it = lane - l_c * (kz + 1);
  const bool l_on = loader && lane < 4 * (kz + 1);
  const int l_cl = (int)cg0 + ((l_on && l_c < ncw) ? l_c : 0);
  const char* const l_base = (l_it == 0) ? reinterpret_cast<const char*>(A.fed_logu + (long long)l_cl * nsteps)
                                         : reinterpret_cast<const char*>(A.fed_z + ((long long)l_cl * nsteps) * kz + (l_it - 1));
  const unsigned int l_stride = (l_it == 0) ? 8u : (unsigned int)(kz * 8);
  const int l_lag = (l_it == 0) ? 1 : 0;                   // step v consumes log-uniform row v - 1 and variate row v
  auto ld_rng = [&](int v) -> double {
    int row = v - l_lag;
    row = row < 0 ? 0 : (row > nsteps - 1 ? nsteps - 1 : row);
    return *reinterpret_cast<const double*>(l_base + (unsigned int)row * l_stride);
  };
  double rng_nx = l_on ? ld_rng(1) : 0.0;   // (staged below, once the increments' mu / scale are known)
  double l_mu = 0.0, l_sc = 1.0;          // item 0 (the log-uniform) passes through: 0 + 1 * x is exact
  if (l_it > 0) {
    int pj = 0, cnt = 0;                  // the (l_it - 1)-th free parameter
    for (int j = 0; j < k; j++)
      if (!A.fixed[j]) { if (cnt == l_it - 1) pj = j; cnt += 1; }
    l_mu = A.mu[pj]; l_sc = A.scale[pj];
  }
  // The tile of step v (log-uniform of v, increments of the proposal v + 1) is staged one whole step ahead -- written in
  // front of barrier v - 1 -- so that every wave reads it BEFORE barrier v, under its in-wave fold, when the LDS pipe is
  // idle.  Behind the barrier all eight waves queue on that pipe and it serves the older waves first: with seven reads
  // per wave there, the partials of the younger waves arrived 410 ticks after the barrier.
  if (l_on) {
    s_rng[1 * 64 + lane] = (l_it > 0) ? l_mu + l_sc * rng_nx : rng_nx;
    rng_nx = ld_rng(2);
  }
  if (tid < 2) s_rng[tid * 64 + 63] = -0.0;
  lds_barrier();
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
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lead)
      A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };

  constexpr bool dbg = DBG;
  unsigned long long te = 0, tb1 = 0, to = 0;   // s_memtime is a ~100-tick scalar-memory round trip: the stamps inside the
  // post-barrier phase are issued without waiting (MFR_STAMP) and collected once at its end
  unsigned long long tsp0 = 0, tsp1 = 0, tsp2 = 0, tsp3 = 0;
// optional stamp inside the phase (-DMFR_STAMP_SEL=4); note that the compiler may move it past pure computations
#ifndef MFR_STAMP_SEL
#define MFR_STAMP_SEL 0
#endif
#define MFR_STAMP(idx, var, dep) unsigned long long var = t_2; if (dbg && MFR_STAMP_SEL == idx) asm volatile("s_memtime %0" : "=s"(var) : "v"(dep))
  for (int v = 1; v <= nsteps; v++) {
    unsigned long long t_0 = dbg ? clk() : 0;
    const int par = v & 1;
    // ================= evaluation of version v of all 4 chains =================
    double wp, lu, dz[HS];
    {
      const double cop = th1[NG];                               // C = intercept of chain j (0 without one)
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t0 = 0; t0 < TN; t0 += MB) {
        {
          constexpr int LAST = TN - 4;             // first pair of the last slot
          const int nu = (TN - t0 < MB) ? TN - t0 : MB;
          double d[MB];
#pragma unroll
          for (int u = 0; u < MB; u++)
            if (u < nu) {
              const int t = t0 + u;
              const double cm = (t >= LAST) ? (((vbits >> (t - LAST)) & 1u) ? cop : 0.0) : cop;
              d[u] = (MFR_X & 32) ? areg[0][t] + cm : __builtin_amdgcn_mfma_f64_4x4x4f64(areg[0][t], th1[0], cm, 0, 0, 0);
            }
#pragma unroll
          for (int q = 1; q < NG; q++)
#pragma unroll
            for (int u = 0; u < MB; u++)
              if (u < nu) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[q][t0 + u], th1[q], d[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < MB; u++)
            if (u < nu) acc[u & 3] = fmh_fma(d[u], d[u], acc[u & 3]);
        }
      }
      // this step's tile: issued here so that the LDS round trip hides under the fold
      lu = s_rng[par * 64 + jch * (kz + 1)];
#pragma unroll
      for (int s = 0; s < HS; s++) dz[s] = s_rng[par * 64 + zix[s]];
      wp = (MFR_X & 4) ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : mfma_wave_fold(acc[0], acc[1], acc[2], acc[3]);
    }
    if (lane < 4) s_wp[(par * 4 + lane) * 8 + wave] = wp;
    if (loader) {   // the tile of step v + 1, fetched one step ago; the next fetch leaves behind the barrier
      if (l_on) s_rng[(par ^ 1) * 64 + lane] = (l_it > 0) ? l_mu + l_sc * rng_nx : rng_nx;
    }
    // sigma-only half of the closed form, in the MFMA slack of the older wave of each SIMD (waves 0..3): wave w publishes,
    // for chain w, n (log sigma + ln sqrt 2 pi), sigma^2, the refined reciprocal of sigma^2 -- the first five
    // instructions of the fp64 division sequence (v_rcp_f64 + two Newton steps), which depend on the denominator only --
    // and sigma itself.  Whatever needs the general path is announced here as well: sigma^2 = 0 and a NaN in the first
    // field (first row, failed chain, sigma outside the fast range), so that the readers test ONE value.
    if (wave < 4 && !(MFR_X & 2)) {
      const double sigma = th1[NG + 1];
      const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
      const bool sg_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;                     // positive, finite, normal
      const double sg = sg_fast ? sigma : 1.0;
      const double t1_fast = fmh_log_pn(sg) + FMH_K(FMH_LN_SQRT_2PI);   // same bits as fmh_log(sigma) on this range
      const double ss = sg * sg;
      const bool okf = sg_fast && mfr_div_safe(ss) && v > 1 && status == FMCMC_CHAIN_OK;
      const double r2 = div_recip(ss);
      if (lane == wave) {
        double* cf = s_cf + (par * 4 + lane) * 6;
        cf[0] = okf ? dn * t1_fast : fmh_nan();
        cf[1] = okf ? ss : 0.0;
        cf[2] = 0.5 * r2;
        cf[3] = r2;
        cf[4] = sigma;
      }
    }
    unsigned long long t_1 = dbg ? clk() : 0;
    lds_barrier();
    unsigned long long t_2 = dbg ? clk() : 0;
    // ================= every lane: fold, decide, propose for its chain =================
    // What bounds this phase is the NUMBER of instructions on the common path, not their dependences: one wave issues
    // a VALU instruction every ~6-7 cycles (tools/dp_latency.hip), and the LDS pipe serves the reads of the older waves
    // first.  Hence: three LDS reads, no selects on masks that never change, nothing computed twice.
    {
      // levels 64, 128, 256 of the canonical tree: lane L fetches the four partials of waves 4 h .. 4 h + 3, h = bit 3 of
      // L (two adds), level 256 is one row_ror:8 exchange
      const double* wsrc = s_wp + (par * 4 + jch) * 8 + 4 * ((lane >> 3) & 1);
      const double w0 = wsrc[0], w1 = wsrc[1], w2 = wsrc[2], w3 = wsrc[3];
      const double* cf = s_cf + (par * 4 + jch) * 6;
      const double nt1_fast = cf[0], ss_fast = cf[1], rinv_h = cf[2], rinv = cf[3];
      if (l_on) rng_nx = ld_rng(v + 2);
      double tot = (w0 + w1) + (w2 + w3);
      if (!(MFR_X & 8)) tot = tot + dpp_d<0x128>(tot);
      // The division (0.5 tot) / sigma^2 is finished with the three instructions of the hardware sequence that depend on
      // the numerator (q0 = h r; e = fma(-d, q0, h); q = fma(e, r, q0)); with the denominator within 2^+-300 and the
      // numerator above 2^-300 v_div_scale / v_div_fmas / v_div_fixup are identities, so these are the bits of `/`
      // (an overflow ends in a NaN and in the general path).  0.5 is folded into the reciprocal: h r == tot (0.5 r).
      const double h = 0.5 * tot;
      const double q0 = tot * rinv_h;
      const double q = fmh_fma(fmh_fma(-ss_fast, q0, h), rinv, q0);
      double f1 = -nt1_fast - q;
      const double ratio_f = f1 - f0;
      // a NaN anywhere (also the publisher's) ends in ratio_f; -inf needs no fix-up (the guard would write -inf again)
      const bool slow = fmh_isnan(ratio_f) || (h < 4.909093465297727e-91);
      bool keep_row = true, acc = false;
      if (__builtin_expect(!__any(slow) || (MFR_X & 16), 1)) {
        acc = (MFR_X & 64) ? false : lu < ratio_f;
      } else {
        // general path, taken by the whole wave as soon as one of its four chains needs it (first row, degenerate
        // sigma, extreme exponents, NaN, failed chains); the same bits as above for the lanes that did not
        keep_row = false;
        if (ss_fast != 0.0 && !(h < 4.909093465297727e-91)) {
          if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
        } else {
          f1 = logpost_of(tot, s_cf[(par * 4 + jch) * 6 + 4]);
        }
        if (v == 1) {
          f0 = f1;
          keep_row = true;
        } else if (status == FMCMC_CHAIN_OK) {
          const double ratio = f1 - f0;
          if (fmh_isnan(f1) || fmh_isnan(ratio)) {
            status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
            if (lead) { A.status[cl] = status; A.status_step[cl] = v; }
#pragma unroll
            for (int s = 0; s < HS; s++)
              if (sten[s]) A.status_theta[(long long)cl * k + pix[s]] = th1[s];
            flush_bits(v);
          } else {
            acc = lu < ratio;
            keep_row = true;
          }
        }
      }
      f0 = acc ? f1 : f0;
#pragma unroll
      for (int s = 0; s < HS; s++) th0[s] = acc ? th1[s] : th0[s];
      MFR_STAMP(4, s_d, th0[0]);
      // waves 0..3: the counters and the rows of chain `wave` (th1 still holds the evaluated proposal)
      if (wave < 4) {
        nacc += acc ? 1 : 0;
        bitword |= (acc ? 1u : 0u) << ((v - 1) & 31);
        if (!(MFR_X & 1) && keep_row && v > burnin) {
          thin_ctr += 1;
          if (thin_ctr == thin) {
            thin_ctr = 0;
#pragma unroll
            for (int s = 0; s < HS; s++)
              if (sten[s]) {
                *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sdoff[s] + srow8)) = th0[s];
                if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sdoff[s] + srow8)) = th1[s];
              }
            if (A.logpost && lead && !dbg) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = f1;
            srow8 += 8;
          }
        }
        if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
      }
      // the next proposal (a failed chain keeps its theta1; the last step proposes into the void)
      if (__builtin_expect(status == FMCMC_CHAIN_OK, 1)) {
#pragma unroll
        for (int s = 0; s < HS; s++) {
          double t = th0[s] + dz[s];
          if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE && zix[s] != 63)
            t = reflect1(t, s_bnd[pix[s] < 0 ? 0 : pix[s]], s_bnd[PIPE_KMAX + (pix[s] < 0 ? 0 : pix[s])]);
          th1[s] = t;
        }
      }
      if (dbg) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tsp3 += s_d - t_2;   // barrier exit -> decision
      }
    }
    if (dbg) {
      unsigned long long t_3 = clk();   // (waits for the stamps above as well)
      te += t_1 - t_0; tb1 += t_2 - t_1; to += t_3 - t_2;
    }
  }
  if (dbg && lane == 0 && A.logpost) {   // stamps leave through the logpost buffer in this diagnostic mode
    double* d = A.logpost + (long long)A.nchains * A.S - 8 * ((long long)blockIdx.x * NW + wave + 1);
    d[0] = (double)te; d[1] = (double)tb1; d[2] = (double)to; d[3] = (double)tsp0; d[4] = (double)nsteps; d[5] = (double)tsp1; d[6] = (double)tsp2; d[7] = (double)tsp3;
  }
  if (wave < 4) {
#pragma unroll
    for (int s = 0; s < HS; s++)
      if (sten[s]) A.theta0[(long long)cl * k + pix[s]] = th0[s];
    if (lead) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    }
  }
}

size_t mfmar_lds_bytes() { return sizeof(double) * (size_t)(MFR_WP + MFR_CF + MFR_RNG + 2 * PIPE_KMAX); }

}  // namespace
