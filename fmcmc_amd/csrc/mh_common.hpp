// mh_common.hpp -- shared device code of the sweep kernels: launch arguments, LDS-only barriers, wave reductions on
// DPP / permlane swaps, reflection, per-chain LDS layout, streamed evaluation and the closed forms of the families.
// Included by mh_engine.hip only (one translation unit; everything lives in its anonymous namespace).
#pragma once

namespace {

constexpr int NT = 512;       // threads per workgroup == canonical lanes
constexpr int NW = NT / 64;   // wavefronts per workgroup
constexpr int MAXK = FMCMC_MAX_K;

thread_local char g_err[1024] = "";
void set_err(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct SweepArgs {
  // model
  int family, p, intercept, guard;
  long long n;
  const double* X;
  const double* y;
  double prior_div;
  // kernel
  int kind, k, scheme, warmup;
  int variate;               // 0: N(0,1) / Student-t by kind; 1: U(0,1) (uniform kernels)
  int freq;                  // ram: adaptation frequency
  int scheme_len;            // explicit scheme
  const int* scheme_seq;     // [scheme_len] 0-based parameter indices (device)
  const double* constr;      // ram: [kf][kf] mask or NULL (device)
  int* scheme_cols;          // [C][nsteps] plan of scheme = "random": in (FED) / out (PHILOX), or NULL
  int nadapt;                // mirror kernels: abs_iter of the one-off scale adaptation
  double* mirror_mu;         // [C][k] in/out
  double* mirror_scale;      // [C][k] in/out
  double* obs_arate;         // [C] out (in when continuing)
  int bw;                    // adapt: window (0 = recursive)
  int hist_rows;             // adapt with bw > 0 / freq > 1: rows of the ring below (max(freq, bw - 1)), else 0
  double Sd;                 // adapt, bw > 0
  double* hist;              // [C][hist_rows][kf] ring of the last rows of ans[, which.] (row r in slot r % hist_rows)
  double until, eps, arate;
  const double* mu;
  const double* scale;
  const double* lb;
  const double* ub;
  const uint8_t* fixed;
  // run
  long long nchains, nsteps, burnin, thin, S, chain_base, step_base;
  unsigned long long seed;
  int rng_mode, fresh, ram_bounded, kz, tb, debug;
  const double* fed_logu;
  const double* fed_z;
  // state
  double* theta0;
  double* f0;
  long long* abs_iter;
  double* Sigma;
  double* mean_prev;
  int* have_mean;
  int* nerrors;
  // out
  double* samples;
  double* logpost;
  double* draws;
  long long* accept_count;
  unsigned int* accept_bits;
  int* status;
  long long* status_step;
  double* status_theta;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Wavefront-level ordering of LDS traffic only.  wave_sync()'s acq_rel fence also orders GLOBAL memory, i.e. it
// waits (vmcnt) for the owner's own row stores and prefetch loads at every one of the dozen sync points of an
// adaptive proposal; the owners only ever exchange data with themselves through LDS.
__device__ __forceinline__ void wave_sync_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also releases GLOBAL memory at
// workgroup scope, i.e. s_waitcnt vmcnt(0): every wave would sit at every barrier until its sample /
// draw / logpost stores (and prefetch loads) have round-tripped.  Nothing a workgroup exchanges
// inside the sweep goes through global memory, so LDS ordering is all the protocol needs.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- wavefront xor-butterfly sum on the DPP / permlane-swap datapath (no LDS crossbar) ------
// Same VALUES as v += shfl_xor(v, 1), 2, 4, 8, 16, 32: after the xor-1/xor-2 steps a quad is
// uniform, so row_half_mirror (lane i <-> 7-i) and row_mirror (i <-> 15-i) deliver exactly the
// partner group's sum; rows / halves are exchanged with v_permlane16_swap / v_permlane32_swap.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_xor_sum(double v) {
  v = v + dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]  : xor 1
  v = v + dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]  : xor 2
  v = v + dpp_d<0x141>(v);  // row_half_mirror      : xor 4 (quads are uniform)
  v = v + dpp_d<0x140>(v);  // row_mirror           : xor 8
  {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    double a = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
    double b = __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    v = a + b;              // xor 16
  }
  {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double a = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
    double b = __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    v = a + b;              // xor 32
  }
  return v;
}

// canonical reflect (twin of oracle reflect1, MATH_CANON branch; R/kernel.R:450-493)
__device__ __forceinline__ double reflect1(double x, double lb, double ub) {
  double d = ub - lb;
  if (x > ub) {
    double e = x - ub;
    double q = e / d, fq = __builtin_floor(q);
    double tmp = fmh_fma(-fq, d, e);
    double q2 = __builtin_floor(tmp / d);
    double dm = fmh_fma(-q2, d, tmp);
    double idiv = fq + q2;
    double odd = idiv - 2.0 * __builtin_floor(0.5 * idiv);
    return (odd != 0.0) ? (lb + dm) : (ub - dm);
  }
  if (x < lb) {
    double e = lb - x;
    double q = e / d, fq = __builtin_floor(q);
    double tmp = fmh_fma(-fq, d, e);
    double q2 = __builtin_floor(tmp / d);
    double dm = fmh_fma(-q2, d, tmp);
    double idiv = fq + q2;
    double odd = idiv - 2.0 * __builtin_floor(0.5 * idiv);
    return (odd != 0.0) ? (ub - dm) : (lb + dm);
  }
  return x;
}

// Per-chain LDS block layout (doubles). LD = kf|1 keeps column walks conflict-free.
struct ChainLds {
  double* th0;   // [k]
  double* th1;   // [k]
  double* vz;    // [kf] z / U
  double* vv;    // [kf] v = S U, or x (adapt)
  double* vmp;   // [kf] mean_prev
  double* vmt;   // [kf] mean_t
  double* vrs;   // [kf] running sum of ans rows (adapt)
  double* SigA;  // [kf*LD]
  double* SigB;  // [kf*LD] adapt: Cholesky factor; ram: the other buffer of S
  double* sc;    // scalars: 0 f0, 1 f1
  double* mmu;   // [k] mirror kernels: adapted mean
  double* msc;   // [k] mirror kernels: adapted scale
};

__host__ __device__ inline int chain_lds_doubles(int k, int kf, int kind) {
  int LD = kf | 1;
  int mats = (kind == FMCMC_KERNEL_ADAPT || kind == FMCMC_KERNEL_RAM) ? 2 * kf * LD : 0;
  int mir = (kind == FMCMC_KERNEL_NMIRROR || kind == FMCMC_KERNEL_UMIRROR) ? 2 * k : 0;
  return 2 * k + 5 * kf + mats + 4 + mir;
}

__device__ __forceinline__ ChainLds chain_lds(double* base, int k, int kf, int kind) {
  ChainLds c;
  int LD = kf | 1;
  c.th0 = base;
  c.th1 = c.th0 + k;
  c.vz = c.th1 + k;
  c.vv = c.vz + kf;
  c.vmp = c.vv + kf;
  c.vmt = c.vmp + kf;
  c.vrs = c.vmt + kf;
  c.SigA = c.vrs + kf;
  int mats = (kind == FMCMC_KERNEL_ADAPT || kind == FMCMC_KERNEL_RAM) ? kf * LD : 0;
  c.SigB = c.SigA + mats;
  c.sc = c.SigB + mats;
  c.mmu = c.sc + 4;
  c.msc = c.mmu + k;
  return c;
}

// ---- workgroup-collective log-posterior partial sums (streamed variant) ------------------
// Every thread accumulates its canonical lane for all CW chains, then the wavefront butterfly
// (levels 1..32) runs and lane 0 of each wavefront publishes its partial to s_part[w][c].
template <int CW>
__device__ __forceinline__ void eval_partials(const SweepArgs& A, double* const* th /*[CW] -> theta in LDS*/,
                                              double* s_part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long n = A.n;
  const int p = (A.family == FMCMC_FAM_IID_NORMAL) ? 0 : A.p;
  const int ic = (A.family == FMCMC_FAM_IID_NORMAL) ? 1 : A.intercept;
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  if (A.family == FMCMC_FAM_LOGISTIC) {
    // The data comes from L2 (~1 us under load) and only two waves share a SIMD: a load-use chain per column made this
    // loop latency-bound (6 dependent round trips per observation).  All columns of an observation are fetched as one
    // batch, and the batch of the NEXT observation is in flight while exp / log1p of the current one run.
    constexpr int JB = 8;
    auto term_of = [&](double e, bool y1) -> double {
      double s = y1 ? e : -e;
      double a = (s < 0.0) ? s : -s;
      double l1 = fmh_log1p_exp_nonpos(a);   // == fmh_log1p(fmh_exp(a)) bit for bit, straight-line on the common range
      return (s < 0.0) ? (s - l1) : (-l1);
    };
    if (p <= JB) {
      double xb[JB], yv = 0.0;
      long long i = tid;
      if (i < n) {
#pragma unroll
        for (int u = 0; u < JB; u++) xb[u] = (u < p) ? A.X[(long long)u * n + i] : 0.0;
        yv = A.y[i];
      }
      for (; i < n; i += NT) {
        double eta[CW];
#pragma unroll
        for (int c = 0; c < CW; c++) eta[c] = ic ? th[c][0] : 0.0;
#pragma unroll
        for (int u = 0; u < JB; u++)
          if (u < p) {
#pragma unroll
            for (int c = 0; c < CW; c++) eta[c] = fmh_fma(xb[u], th[c][ic + u], eta[c]);
          }
        const bool y1 = (yv != 0.0);
        const long long inx = (i + NT < n) ? i + NT : i;   // clamped: the last prefetch re-reads this observation
#pragma unroll
        for (int u = 0; u < JB; u++) xb[u] = (u < p) ? A.X[(long long)u * n + inx] : 0.0;
        yv = A.y[inx];
#pragma unroll
        for (int c = 0; c < CW; c++) acc[c] = acc[c] + term_of(eta[c], y1);
      }
    } else {
      for (long long i = tid; i < n; i += NT) {
        double eta[CW];
#pragma unroll
        for (int c = 0; c < CW; c++) eta[c] = ic ? th[c][0] : 0.0;
        const bool y1 = (A.y[i] != 0.0);
        int j = 0;
        for (; j + JB <= p; j += JB) {
          double xb[JB];
#pragma unroll
          for (int u = 0; u < JB; u++) xb[u] = A.X[(long long)(j + u) * n + i];
#pragma unroll
          for (int u = 0; u < JB; u++) {
#pragma unroll
            for (int c = 0; c < CW; c++) eta[c] = fmh_fma(xb[u], th[c][ic + j + u], eta[c]);
          }
        }
        for (; j < p; j++) {
          double x = A.X[(long long)j * n + i];
#pragma unroll
          for (int c = 0; c < CW; c++) eta[c] = fmh_fma(x, th[c][ic + j], eta[c]);
        }
#pragma unroll
        for (int c = 0; c < CW; c++) acc[c] = acc[c] + term_of(eta[c], y1);
      }
    }
  } else {
    // Memory-level parallelism: the data comes from L2 (latency ~1 us under load), so every thread keeps a
    // batch of JB independent column loads in flight before the FMAs that consume them; a dependent
    // load-use chain per (observation, column) left < 16 KB in flight per CU (10x below the L2 rate at k = 50).
    // Measured (tools/bench_cw.py, k = 50, 512 chains): 2 chains per workgroup is the optimum (59 us per step; 1: 114,
    // 4: 70, 8: 120), i.e. the loop is bound by load latency + FMA issue per CU (~65 GB/s per CU of the 154 GB/s L1 fill
    // rate), not by aggregate L2 bandwidth; a second batch in flight (double-buffered xb) spills in this all-kinds kernel
    // and is 12 % slower.
    constexpr int JB = 8;
    for (long long i = tid; i < n; i += NT) {
      double mu[CW];
#pragma unroll
      for (int c = 0; c < CW; c++) mu[c] = ic ? th[c][0] : 0.0;
      const double yv = A.y[i];
      int j = 0;
      for (; j + JB <= p; j += JB) {
        double xb[JB];
#pragma unroll
        for (int u = 0; u < JB; u++) xb[u] = A.X[(long long)(j + u) * n + i];
#pragma unroll
        for (int u = 0; u < JB; u++) {
#pragma unroll
          for (int c = 0; c < CW; c++) mu[c] = fmh_fma(xb[u], th[c][ic + j + u], mu[c]);
        }
      }
      for (; j < p; j++) {
        double x = A.X[(long long)j * n + i];
#pragma unroll
        for (int c = 0; c < CW; c++) mu[c] = fmh_fma(x, th[c][ic + j], mu[c]);
      }
#pragma unroll
      for (int c = 0; c < CW; c++) {
        double r = yv - mu[c];
        acc[c] = fmh_fma(r, r, acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CW; c++) {
    double v = wave_xor_sum(acc[c]);
    if (lane == 0) s_part[wave * CW + c] = v;
  }
}

// closed form of the family given the canonical total `tot`. Uniform over the wave.
__device__ __forceinline__ double finish_logpost(const SweepArgs& A, const double* th, double tot) {
  double f;
  if (A.family == FMCMC_FAM_LOGISTIC) {
    f = tot;
    if (A.prior_div != 0.0) {
      double ss = 0.0;
      const int nb = A.intercept + A.p;
      for (int j = 0; j < nb; j++) ss = fmh_fma(th[j], th[j], ss);
      f = f - ss / A.prior_div;
    }
  } else {
    const int pp = (A.family == FMCMC_FAM_IID_NORMAL) ? 0 : A.p;
    const int ic = (A.family == FMCMC_FAM_IID_NORMAL) ? 1 : A.intercept;
    const double sigma = th[ic + pp];
    if (sigma < 0.0 || fmh_isnan(sigma)) {
      f = fmh_nan();
    } else if (sigma == 0.0) {
      f = -fmh_inf();
    } else {
      double t1 = fmh_log(sigma) + FMH_LN_SQRT_2PI;
      double q = (0.5 * tot) / (sigma * sigma);
      f = -((double)A.n * t1) - q;
    }
  }
  if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
  return f;
}

}  // namespace
