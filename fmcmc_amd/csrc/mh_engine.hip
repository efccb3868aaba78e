// mh_engine.hip — gfx950 (MI355X) many-chain Metropolis-Hastings sweep kernels + C-ABI.
//
// Replaces, for ALL chains of a call at once, the per-chain loop of the reference
//   R/mcmc.R:720-838 (loop, accept, burn-in/thin)  x  R/kernel_normal.R / R/kernel_adapt.R /
//   R/kernel_ram.R / R/recursive.R / R/kernel.R:450-493 (proposal kernels).
//
// Execution model (DESIGN.md has the full picture):
//   * one 512-thread workgroup (8 wavefronts) owns CW chains; its 512 threads ARE the 512
//     "canonical lanes" of the log-posterior reduction: observation i belongs to lane i mod 512,
//     each lane accumulates its observations in index order with fma, lanes are combined by an
//     xor-butterfly tree (levels 1..32 inside a wavefront, 64..256 across the 8 wavefronts).
//     The CPU oracle mirrors exactly this tree, so accept decisions are bit-identical.
//   * every data value loaded by a thread is applied to all CW chains of the workgroup
//     (register/L2 traffic amortised over chains);
//   * per-chain "scalar" work (proposal, adaptation, accept) is done by the chain's owner
//     wavefront, lanes = parameters (one lane per row of Sigma / S);
//   * RNG = Philox4x32-10 counter stream (include/fmh_philox.h) or host-fed variates.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fma only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>
#include <vector>

#include "../../include/fmcmc_amd.h"
#include "../../include/fmh_detmath.h"
#include "../../include/fmh_philox.h"

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

// ---- the sweep kernel ----------------------------------------------------------------------
// P < 0 : streamed evaluation (any family, any n, p: data re-read from L2 every step)
// P >= 0: register-resident Gaussian linear regression with P covariates: each thread keeps its
//         OPT observations (x[P], y) in VGPRs for the whole sweep; n in (512*(OPT-4), 512*OPT].
constexpr int RES_MASKED = 4;  // trailing observation slots that carry a validity mask

// KIND > 0 compiles exactly one proposal kernel in (resident variants); KIND == 0 keeps all four
// behind the runtime A.kind (streamed variants).
template <int CW, int P, int OPT, int KIND>
__global__ __launch_bounds__(NT) void mh_sweep_kernel(const SweepArgs A0) {
  constexpr bool RESIDENT = (P >= 0);
  SweepArgs A = A0;
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
  int* s_which = (int*)(s_ub + k);          // [k] ints (k/2+1 doubles)
  double* s_part = s_ub + k + (k / 2 + 1);  // [NW*CW]
  int* s_flag = (int*)(s_part + NW * CW);   // [2] ints
  double* s_zt = s_part + NW * CW + 1;      // [CW][TB][kz] proposal variates of the tile
  double* s_lu = s_zt + CW * TB * kz;       // [CW][TB]     log accept uniforms of the tile
  double* s_tr = s_lu + CW * TB;            // RESIDENT: [CW][NT] lane partials
  double* s_chains = s_tr + (RESIDENT ? CW * NT : 0);

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
  }
  __syncthreads();
  const int kf = s_kf;
  const int LD = kf | 1;
  const int CHS = chain_lds_doubles(k, kf, A.kind);
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
      eval_partials<CW>(A, thp, s_part);
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
  double obs_arate = fmh_nan();   // mirror kernels
  long long nzero = 0;            // rows 2..i-1 of this call equal to their predecessor (rowSums(diff(ans)^2) == 0)
  double* Scur = L.SigA;   // ram: current factor buffer
  double* Salt = L.SigB;

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
      if (!A.fresh) { abs_iter = A.abs_iter[cl]; obs_arate = A.obs_arate[cl]; }
    }
    if (adaptive) {
      if (A.fresh) {
        for (int e = lane; e < kf * LD; e += 64) {
          int a = e / LD, b = e % LD;
          L.SigA[e] = (a == b) ? 1.0 * A.eps : 0.0;
          L.SigB[e] = 0.0;
        }
      } else {
        for (int e = lane; e < kf * LD; e += 64) {
          int a = e / LD, b = e % LD;
          L.SigA[e] = (b < kf) ? A.Sigma[(cl * kf + a) * kf + b] : 0.0;
          L.SigB[e] = 0.0;
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
  double* const out_s = A.samples + (cl * k + (lane < k ? lane : 0)) * S;
  double* const out_d = A.draws ? A.draws + (cl * k + (lane < k ? lane : 0)) * S : nullptr;
  double* const out_l = A.logpost ? A.logpost + cl * S : nullptr;
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
    f0 = finish_logpost(A, L.th1, total_of(myc));
    f1 = f0;
    if (lane < kf) L.vrs[lane] = L.th0[s_which[lane]];
    if (A.hist_rows > 0 && lane < kf) A.hist[((long long)cl * A.hist_rows + (1 % A.hist_rows)) * kf + lane] = L.th0[s_which[lane]];
    store_row(1, f0);
  }

  // ---- main loop
  int tt = -1;         // position inside the RNG tile
  int ord = 0;         // ordered scheme: (i - 1) mod kf
  for (int i = 2; i <= nsteps; i++) {
    tt = (tt + 1 == TB) ? 0 : tt + 1;
    ord = (ord + 1 == kf) ? 0 : ord + 1;
    bool ram_gate = false;
    // ================= RNG tile: all 512 threads draw the variates of the next TB steps ===========
    if (tt == 0) {
      __syncthreads();  // owners are done with the previous tile (and with s_part)
      const int per_c = TB * (kz + 1);
      for (int idx = tid; idx < CW * per_c; idx += NT) {
        const int c = idx / per_c, rem = idx - c * per_c;
        const int t = rem / (kz + 1), a = rem - t * (kz + 1);
        const long long ii = (long long)i + t;
        if (c < ncw && ii <= A.nsteps) {
          const long long clc = cg0 + c;
          const unsigned int cg = (unsigned int)(A.chain_base + clc);
          const unsigned int st = (unsigned int)(A.step_base + ii);
          double v;
          if (a == kz) {
            v = (A.rng_mode == FMCMC_RNG_FED) ? A.fed_logu[clc * A.nsteps + (ii - 1)] : fmh_log_accept_u(A.seed, st, cg);
            s_lu[c * TB + t] = v;
          } else {
            if (A.rng_mode == FMCMC_RNG_FED) v = A.fed_z[(clc * A.nsteps + (ii - 1)) * kz + a];
            else if (A.kind == FMCMC_KERNEL_RAM) v = fmh_student_t(A.seed, st, cg, (unsigned int)a, (double)kf);
            else if (A.variate == 1) v = fmh_unif(A.seed, st, cg, (unsigned int)a);
            else v = fmh_normal(A.seed, st, cg, (unsigned int)a);
            s_zt[(c * TB + t) * kz + a] = v;
          }
        }
      }
      __syncthreads();
    }
    const double* zt = s_zt + ((owner ? myc : 0) * TB + tt) * kz;
    // ================= scalar phase A: proposal =================
    if (owner && status == FMCMC_CHAIN_OK) {
      if (A.kind == FMCMC_KERNEL_NORMAL || A.kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) {
        if (lane < k) L.th1[lane] = L.th0[lane];
        wave_sync();
        const bool refl = (A.kind == FMCMC_KERNEL_NORMAL_REFLECTIVE);
        // plan_update_sequence (R/kernel.R:66-133): every scheme but "joint" updates ONE parameter per step
        const bool single = (A.scheme != FMCMC_SCHEME_JOINT);
        int col = 0;
        if (A.scheme == FMCMC_SCHEME_ORDERED) {
          col = s_which[ord];
        } else if (A.scheme == FMCMC_SCHEME_EXPLICIT) {
          col = A.scheme_seq[(i - 1) % A.scheme_len];
        } else if (A.scheme == FMCMC_SCHEME_RANDOM) {
          if (A.rng_mode == FMCMC_RNG_FED) {
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
        }
        if (lane < k) L.th1[lane] = L.th0[lane];
        wave_sync();
        const bool single = (A.scheme != FMCMC_SCHEME_JOINT);
        int col = 0;
        if (A.scheme == FMCMC_SCHEME_ORDERED) {
          col = s_which[ord];
        } else if (A.scheme == FMCMC_SCHEME_EXPLICIT) {
          col = A.scheme_seq[(i - 1) % A.scheme_len];
        } else if (A.scheme == FMCMC_SCHEME_RANDOM) {
          if (A.rng_mode == FMCMC_RNG_FED) {
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
          if (H > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // the ring rows were stored by other lanes of this wave
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
                wave_sync();
                if (lane < kf)
                  for (int b = 0; b < kf; b++) L.SigA[lane * LD + b] = fmh_fma(d, L.vv[b], L.SigA[lane * LD + b]);
                wave_sync();
              }
              if (lane < kf)
                for (int b = 0; b < kf; b++) {
                  const double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                  L.SigA[lane * LD + b] = A.Sd * (L.SigA[lane * LD + b] / (double)(N - 1) + ik);
                }
              wave_sync();
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
              wave_sync();
              if (lane < kf) {
                const double c1 = (t - 1) / t, c2 = 1.0 / t;
                for (int b = 0; b < kf; b++) {
                  double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                  double inner = t * (mp * L.vmp[b]) - (t + 1) * (mt * L.vmt[b]) + x * L.vv[b] + 1e-5 * ik;
                  L.SigA[lane * LD + b] = c1 * L.SigA[lane * LD + b] + c2 * inner;
                }
              }
              wave_sync();
              if (lane < kf) L.vmp[lane] = mt;
              have_mean = 1;
            }
          }
        }
        abs_iter += 1;
        // left-looking Cholesky, lane = row (twin of oracle chol_lower_canon)
        bool notpd = false;
        for (int j = 0; j < kf && status == FMCMC_CHAIN_OK; j++) {
          double s = 0.0;
          if (lane >= j && lane < kf) {
            s = L.SigA[lane * LD + j];
            for (int b = 0; b < j; b++) s = fmh_fma(-L.SigB[lane * LD + b], L.SigB[j * LD + b], s);
          }
          double d = shfl_d(s, j);
          if (!(d > 0.0) || !fmh_isfinite(d)) { notpd = true; break; }
          double ljj = fmh_sqrt(d);
          if (lane == j) L.SigB[j * LD + j] = ljj;
          else if (lane > j && lane < kf) L.SigB[lane * LD + j] = s / ljj;
          wave_sync();
        }
        if (notpd) {
          status = FMCMC_CHAIN_NOT_PD;
        } else if (status == FMCMC_CHAIN_OK) {
          if (lane < k) L.th1[lane] = L.th0[lane];
          wave_sync();
          if (lane < kf) {
            double s = 0.0;
            for (int b = 0; b <= lane; b++) s = fmh_fma(L.SigB[lane * LD + b], zt[b], s);
            int j = s_which[lane];
            double t = L.th0[j] + (s_mu[j] + s);
            L.th1[j] = reflect1(t, s_lb[j], s_ub[j]);
          }
        }
      } else {  // RAM, R/kernel_ram.R:123-126
        if (lane < kf) {
          double s = 0.0;
          for (int b = 0; b <= lane; b++) s = fmh_fma(Scur[lane * LD + b], zt[b], s);
          L.vv[lane] = s;
          int j = s_which[lane];
          L.th1[j] = L.th0[j] + s;
        }
        ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && (i % A.freq) == 0);
      }
      if (status != FMCMC_CHAIN_OK) {  // raised inside the proposal (NOT_PD)
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (lane < k) A.status_theta[cl * k + lane] = L.th1[lane];
      }
    }
    __syncthreads();
    // ================= collective evaluation of f(theta1) =================
    evaluate();
    __syncthreads();
    // ================= scalar phase B: RAM adaptation (needs f(theta1) un-reflected) =================
    if (A.kind == FMCMC_KERNEL_RAM) {
      bool changed = false;
      if (owner && status == FMCMC_CHAIN_OK) {
        if (ram_gate) {
          double f1u = finish_logpost(A, L.th1, total_of(myc));
          double a_n = fmh_exp(f1u - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          double eta = (double)kf * fmh_exp((-2.0 / 3.0) * fmh_log((double)i));
          if (eta > 1.0) eta = 1.0;
          double nrm2 = 0.0;
          for (int b = 0; b < kf; b++) nrm2 = fmh_fma(zt[b], zt[b], nrm2);
          double cp = (eta * (a_n - A.arate)) / nrm2;
          if (cp != 0.0 && fmh_isfinite(cp)) {
            const bool up = cp > 0.0;
            const double scl = fmh_sqrt(fmh_abs(cp));
            double w = (lane < kf) ? scl * L.vv[lane] : 0.0;
            bool fail = false;
            for (int j = 0; j < kf; j++) {
              double ljj = Scur[j * LD + j];
              double xj = shfl_d(w, j);
              double r2 = up ? fmh_fma(xj, xj, ljj * ljj) : fmh_fma(-xj, xj, ljj * ljj);
              if (!(r2 > 0.0) || !fmh_isfinite(r2)) { fail = true; break; }
              double r = fmh_sqrt(r2);
              double cc = r / ljj, ss = xj / ljj;
              if (lane == j) {
                Salt[j * LD + j] = r;
              } else if (lane > j && lane < kf) {
                double lij = Scur[lane * LD + j];
                double ln = (up ? fmh_fma(ss, w, lij) : fmh_fma(-ss, w, lij)) / cc;
                w = fmh_fma(-ss, ln, cc * w);
                Salt[lane * LD + j] = ln;
              }
            }
            wave_sync();
            if (fail) {
              nerr += 1;
            } else {
              double* t = Scur; Scur = Salt; Salt = t;
            }
          }
          if (A.constr) {  // Sigma <<- constr[which., which.] * Sigma (R/kernel_ram.R:149-150)
            if (lane < kf)
              for (int b = 0; b < kf; b++) Scur[lane * LD + b] = A.constr[lane * kf + b] * Scur[lane * LD + b];
            wave_sync();
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
    // ================= scalar phase C: accept / store (R/mcmc.R:754-778) =================
    if (owner && status == FMCMC_CHAIN_OK) {
      f1 = finish_logpost(A, L.th1, total_of(myc));
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (lane < k) A.status_theta[cl * k + lane] = L.th1[lane];
      } else {
        const double lu = s_lu[myc * TB + tt];
        bool moved = false;
        if (lu < ratio) {
          if (mirror) {   // rowSums(diff(ans)^2) of the row about to be stored (sequential sum, as in the oracle)
            double sq = 0.0;
            for (int a = 0; a < k; a++) sq = sq + (L.th1[a] - L.th0[a]) * (L.th1[a] - L.th0[a]);
            moved = (sq != 0.0);
            wave_sync();
          }
          if (lane < k) L.th0[lane] = L.th1[lane];
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        if (mirror && !moved) nzero += 1;
        wave_sync();
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
  }

  // ---- write state back
  if (owner) {
    if (lane < k) A.theta0[cl * k + lane] = L.th0[lane];
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
      if (mirror) { A.abs_iter[cl] = abs_iter; A.obs_arate[cl] = obs_arate; }
      if (adaptive) {
        A.abs_iter[cl] = abs_iter;
        if (A.nerrors) A.nerrors[cl] = nerr;
        if (A.kind == FMCMC_KERNEL_ADAPT) A.have_mean[cl] = have_mean;
      }
    }
    if (mirror && lane < k) {
      A.mirror_mu[cl * k + lane] = L.mmu[lane];
      A.mirror_scale[cl * k + lane] = L.msc[lane];
    }
    if (adaptive) {
      wave_sync();
      const double* Sfin = (A.kind == FMCMC_KERNEL_RAM) ? Scur : L.SigA;
      for (int e = lane; e < kf * kf; e += 64) {
        int a = e / kf, b = e % kf;
        A.Sigma[(cl * kf + a) * kf + b] = Sfin[a * LD + b];
      }
      if (A.kind == FMCMC_KERNEL_ADAPT && lane < kf) A.mean_prev[cl * kf + lane] = L.vmp[lane];
    }
  }
}

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
                                                       long long nsteps, int kz, int student_df,
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
  if (student_df > 0) {  // kernel_ram: qfun = rt(k, k)
    for (int a = 0; a < kz; a++) z[item * kz + a] = fmh_student_t(seed, st, cg, (unsigned int)a, (double)student_df);
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
__device__ __forceinline__ double uniform_d(double v) {  // pin a wave-uniform double into SGPRs
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double readlane_d(double v, int src) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
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
  const unsigned int sd_off = (unsigned int)((((long long)cl * k + jl) * A.S) * 8);       // samples / draws column
  const unsigned int z_off = (unsigned int)((((long long)cl * nsteps) * kz + zidx) * 8);  // this lane's z column
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.S) * 8);
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

// ==============================================================================================
// Wave-specialised resident kernel (the headline path).
//
//   768 threads = 12 wavefronts per workgroup, 3 per SIMD:
//     waves 0..7  COMPUTE: hold the x columns of their 64 canonical lanes in VGPRs (y in LDS) and do nothing
//                 but evaluate: for version v, for chain c: wait ready[c] >= v, read theta1[c], 20 observations
//                 x (3 fma + sub + fma), write the lane partial, arrive on done[c].
//     waves 8..11 OWNERS (one per chain): wait done[c] == 8 v, fold the 512 partials (canonical tree), closed
//                 form, accept, propose, prefetch, publish theta1[c] (ready[c] = v + 1), then store the row.
//   No s_barrier in the steady state: producers/consumers meet on LDS sequence words, so an owner's
//   latency-bound phase overlaps the evaluation of the OTHER three chains, and on every SIMD the owner's
//   dependency stalls are filled by the two compute waves' independent FMAs (hardware multithreading instead
//   of compiler interleaving).  Register budget: 12 waves -> 168 VGPRs; one chain per evaluation pass keeps the
//   compute role at 120 (data) + ~30: 148 VGPRs, no scratch.
//   FMCMC_AMD_DEBUG_MODE=8 stamps (s_memtime) flag-wait / work time per wave into the draws buffer.
// ==============================================================================================
constexpr int SPEC_NT = 768;
constexpr int SPEC_NCW = 8;   // compute wavefronts

__device__ __forceinline__ unsigned lds_ld_u32(const unsigned* p) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}

__device__ __forceinline__ double lds_ld_f64(const double* p) {   // ordered after a preceding flag poll
  double v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}

// ==============================================================================================
// MFMA evaluation kernel for the headline shape: Gaussian linear regression with exactly 3 covariates.
//
// v_mfma_f64_4x4x4_4b_f64 on gfx950 (measured, tools/mfma64_exact.hip): k = lane / 16; inside a 16-lane group the
// A operand sits at 4*blk + i, B at 4*blk + j, the result D at lane 16*i + 4*blk + j; and it is BITWISE the chain
//     D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))).
// With A = [x1, x2, x3, y], B = [b1, b2, b3, -1] and C = b0 this is exactly the canonical
//     m = fma(x3, b3, fma(x2, b2, fma(x1, b1, b0)));   -r = fma(y, -1, m)
// for 16 observations x 4 chains per instruction, and fma(D, D, acc) == fma(r, r, acc) bit for bit.
// One MFMA replaces 16 x 4 x 4 = 256 VALU lane-FMAs + 64 subtractions with ONE issue slot, which lifts the
// fp64 pipe out of the VALU issue limit (~6.2 ticks per instruction at 2 waves per SIMD).
//
// Mapping that keeps the canonical reduction: wave w owns canonical lanes 64w..64w+63; MFMA t = 4*s + g covers
// slot s (observations i = lane + 512 s) of lanes 64w + 16g + o, o = 4*blk + i_row; result lane L holds chain
// j = L % 4 of canonical lane 64w + 16g + 4*((L/4)%4) + L/16, accumulated in acc[g] in slot order; the four
// accumulators are written to the same transposed partial tile the owners fold with the canonical tree.
// All four chains of the workgroup are evaluated together, so this kernel is not chain-pipelined: a step is
// evaluation | barrier | 4 owner phases in parallel (waves 0..3, priority raised) | barrier.
// ==============================================================================================
constexpr int MF_NMF = 80;   // MFMAs per wave per step: 20 slots x 4 lane groups
// chain stride of the partial tiles: here one ds_write_b64 carries 4 chains x 16 canonical lanes; with the row stride
// 66 the 16 lanes of a write group land on double-banks {0,8,1,9} + 2*(l&7), so a chain stride == 2 (mod 16) spreads the
// four chains over all 16 double-banks (8*66 = 528 == 0 mod 16 made every write a 4-way conflict)
constexpr int MF_TCS = 8 * PIPE_TRS + 2;

// DBG is a TEMPLATE parameter on purpose: the MFMA loop is sensitive to every live register (fewer free VGPRs = fewer
// MFMA results in flight before their dependent fma); stamp code that is merely disabled at run time cost 13 %.
// Shapes: NG groups of K = 4 carry up to 4*NG - 1 covariates plus y (unused k-slots are zero: fma(0, 0, acc) == acc
// exactly), the 80 operand registers of a lane hold 20 / NG observation slots, i.e. NG = 1: p <= 3, n <= 10240;
// NG = 2: p <= 7, n <= 5120.  The number of observation slots NS = ceil(n / 512) is a TEMPLATE parameter: with a run-time
// count every batch of MFMAs becomes a basic block, the scheduler can no longer overlap the FMAs of one batch with the
// MFMAs of the next, and the step is 22 % slower (measured).  Only the last slot holds padding and pays for masks.
template <int KIND, int NG, int NS, bool DBG>
__global__ __launch_bounds__(NT) void mh_sweep_mfma(const SweepArgs A) {
  constexpr int CW = 4;
  constexpr int MB = 8;             // (slot, lane group) pairs per batch = 2 observation slots
  constexpr int TN = NS * 4;        // pairs held per lane and group (NG * TN <= MF_NMF registers)
  static_assert(NG * TN <= MF_NMF, "operand registers");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX]
  double* s_par = s_th1 + CW * PIPE_KMAX;          // [4][PIPE_KMAX]
  double* s_tr = s_par + 4 * PIPE_KMAX;            // [CW][MF_TCS] transposed partial tiles, chain stride 530
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int ic = A.intercept;

  // ---- A operand: feature kf_ = lane / 16 of observation (64 w + 16 g + lane % 16) + 512 s, for t = 4 s + g
  const int feat = lane >> 4, o16 = lane & 15;
  const int P = A.p;
  double areg[NG][TN];
#pragma unroll
  for (int q = 0; q < NG; q++) {
    const int f = 4 * q + feat;   // column of [x_1 .. x_P, y, 0 ..] this lane feeds as operand A of group q
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
  // result lane L: chain j = L % 4, canonical lane 64 w + 16 g + 4*((L/4)%4) + L/16
  const int jch = lane & 3;
  const int cl_in_g = 4 * ((lane >> 2) & 3) + (lane >> 4);
  unsigned vbits = 0;    // validity of this lane's 4 results in the LAST slot (all earlier slots are full)
  int trs[4];            // transposed tile slot of this lane's canonical lane, per group
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int l = 64 * wave + 16 * g + cl_in_g;
    trs[g] = (l & 7) * PIPE_TRS + (l >> 3);
    if ((long long)l + (long long)NT * (NS - 1) < A.n) vbits |= 1u << g;
  }
  if (tid < k) {
    s_par[0 * PIPE_KMAX + tid] = A.mu[tid];
    s_par[1 * PIPE_KMAX + tid] = A.scale[tid];
    s_par[2 * PIPE_KMAX + tid] = A.lb[tid];
    s_par[3 * PIPE_KMAX + tid] = A.ub[tid];
  }
  if (tid < CW * PIPE_KMAX) {
    const int c = tid / PIPE_KMAX, j = tid - c * PIPE_KMAX;
    s_th1[tid] = (c < ncw && j < k) ? A.theta0[(cg0 + c) * k + j] : 0.0;
  }

  // ---- owner state (waves 0..3), as in mh_sweep_spec
  const int myc = wave;
  const bool owner = (myc < ncw);
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + (owner ? myc : 0));
  const bool plane = owner && (lane < k);
  const int jl = (lane < k) ? lane : 0;
  const bool fixed_l = A.fixed[jl] != 0;
  int zidx = 0;
  for (int j = 0; j < jl; j++) zidx += A.fixed[j] ? 0 : 1;
  if (zidx > kz - 1) zidx = kz > 0 ? kz - 1 : 0;   // lanes without a variate of their own read a valid neighbour (value unused)
  double th0 = plane ? A.theta0[(long long)cl * k + lane] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0;
  unsigned int srow8 = 0, bitword = 0;
  const unsigned int sd_off = (unsigned int)((((long long)cl * k + jl) * A.S) * 8);
  const unsigned int z_off = (unsigned int)((((long long)cl * nsteps) * kz + zidx) * 8);
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.S) * 8);
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.fed_z) + (z_off + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  // every lane of an owner wavefront keeps one variate and the log-uniform of the next step in flight (unconditional,
  // clamped addresses: a conditional load costs a register copy behind the load, i.e. an exposed wait)
  double z_nx = (owner && kz > 0) ? ld_z(nsteps >= 2 ? 1 : 0) : 0.0;
  double lu_nx = owner ? lu_row[nsteps >= 2 ? 1 : 0] : 0.0;
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
    if (A.accept_bits && lane == 0)
      A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };
  lds_barrier();

  constexpr bool dbg = DBG;
  bool st_keep = false;                      // row of the step just decided, stored after the barrier
  double st_th0 = 0.0, st_th1 = 0.0, st_f1 = 0.0;
  unsigned long long te = 0, tb1 = 0, to = 0, tb2 = 0, tf = 0, tc = 0, td = 0;
  for (int v = 1; v <= nsteps; v++) {
    unsigned long long t_0 = dbg ? clk() : 0;
    // ================= evaluation of version v of all 4 chains =================
    {
      const double* tj = s_th1 + jch * PIPE_KMAX;
      double bop[NG];                                            // B[k][blk][j] of group q
#pragma unroll
      for (int q = 0; q < NG; q++) {
        const int f = 4 * q + feat;
        bop[q] = (f < P) ? tj[ic + f] : (f == P ? -1.0 : 0.0);
      }
      const double cop = ic ? tj[0] : 0.0;                      // C = intercept of chain j
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      // batches of MB independent MFMA chains followed by their MB dependent FMAs: the result latency of one MFMA is
      // covered by issuing the next ones, and the batch shape (not the allocator's leftovers) bounds the live results
#pragma unroll
      for (int t0 = 0; t0 < TN; t0 += MB) {
        {
          constexpr int LAST = TN - 4;             // first pair of the last slot
          const int nu = (TN - t0 < MB) ? TN - t0 : MB;
          double d[MB];
          // Only the last slot has padding.  A padded observation has A = 0 in every group, so its result is the C
          // operand: feeding 0 instead of the intercept there makes -r == 0 exactly, and the accumulation below is the
          // same straight-line code for every batch (masking the RESULTS put selects in front of the last FMAs).
#pragma unroll
          for (int u = 0; u < MB; u++)
            if (u < nu) {
              const int t = t0 + u;
              const double cm = (t >= LAST) ? (((vbits >> (t - LAST)) & 1u) ? cop : 0.0) : cop;
              d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[0][t], bop[0], cm, 0, 0, 0);
            }
#pragma unroll
          for (int q = 1; q < NG; q++)
#pragma unroll
            for (int u = 0; u < MB; u++)
              if (u < nu) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[q][t0 + u], bop[q], d[u], 0, 0, 0);
          // d = -r of 16 observations x 4 chains per pair
#pragma unroll
          for (int u = 0; u < MB; u++)
            if (u < nu) acc[u & 3] = fmh_fma(d[u], d[u], acc[u & 3]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; g++) s_tr[jch * (MF_TCS) + trs[g]] = acc[g];
    }
    // (Measured: moving log(sigma) of the owners in front of, or right behind, their MFMAs makes the step SLOWER:
    //  fp64 VALU work issued while the SIMD partner runs MFMAs slows those -- one fp64 datapath -- whereas in
    //  the owner phase below that datapath is idle.)
    // sigma-only part of the closed form: the owner waves are the first of their SIMD to finish their MFMAs (the older
    // wave wins the arbitration) and would wait ~1200 ticks at the barrier; its ~65 fp64 instructions run there, in the
    // shadow of the partner wave's MFMAs, instead of in the exposed owner phase.  The results are wave-uniform (SGPRs).
    double sigma = 0.0, nt1_fast = 0.0, ss_fast = 1.0;
    bool sg_fast = false;
    if (owner) {
      sigma = readlane_d(th1, k - 1);
      const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
      sg_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;                     // positive, finite, normal
      const double sg = sg_fast ? sigma : 1.0;
      const double t1_fast = fmh_log_pn(sg) + FMH_K(FMH_LN_SQRT_2PI);   // same bits as fmh_log(sigma) on this range
      nt1_fast = uniform_d(dn * t1_fast);
      ss_fast = uniform_d(sg * sg);
    }
    unsigned long long t_1 = dbg ? clk() : 0;
    lds_barrier();
    unsigned long long t_2 = dbg ? clk() : 0;
    // ================= owners: fold, decide, propose =================
    if (owner) {
      __builtin_amdgcn_s_setprio(3);
      const double* src = s_tr + myc * (MF_TCS) + lane;
      const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
      const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
      // increment of the NEXT proposal: consumes the variate fetched one step ago and refills the same register at once, so
      // that load has a whole step to land and no vector-memory wait sits behind the decision below
      // The only vector-memory wait of the phase sits HERE, on loads issued one whole step ago; both registers are refilled
      // at once, so nothing younger than an evaluation is ever waited for (a wait behind the decision would also cover the
      // refill of the variate, an HBM miss every third step).
      double lu = lu_nx, zc = z_nx;
      asm volatile("" : "+v"(lu), "+v"(zc));
      lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
      if (kz > 0) z_nx = ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1);
      const double dz = (plane && !fixed_l) ? s_par[0 * PIPE_KMAX + lane] + s_par[1 * PIPE_KMAX + lane] * zc : 0.0;
      const double tot = wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
      unsigned long long t_a = dbg ? clk() : 0;
      double f1;
      if (sg_fast) {
        f1 = -nt1_fast - (0.5 * tot) / ss_fast;
        if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
      } else {
        f1 = logpost_of(tot, sigma);
      }
      unsigned long long t_b = dbg ? clk() : 0;
      const double th1_eval = th1;
      bool keep_row = false;
      if (v == 1) {
        f0 = uniform_d(f1);
        keep_row = true;
      } else if (status == FMCMC_CHAIN_OK) {
        const double ratio = f1 - f0;
        if (fmh_isnan(f1) || fmh_isnan(ratio)) {
          status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
          if (lane == 0) { A.status[cl] = status; A.status_step[cl] = v; }
          if (plane) A.status_theta[(long long)cl * k + lane] = th1;
          flush_bits(v);
        } else {
          if (lu < ratio) {
            th0 = th1;
            f0 = uniform_d(f1);
            nacc += 1;
            bitword |= (1u << ((v - 1) & 31));
          }
          keep_row = true;
        }
      }
      unsigned long long t_c = dbg ? clk() : 0;
      if (dbg) { tf += t_a - t_2; tc += t_b - t_a; td += t_c - t_b; }
      const double th0_row = th0;
      if (v < nsteps && status == FMCMC_CHAIN_OK && plane) {
        double t = th0;
        if (!fixed_l) {
          t = th0 + dz;
          if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) t = reflect1(t, s_par[2 * PIPE_KMAX + lane], s_par[3 * PIPE_KMAX + lane]);
        }
        th1 = t;
        s_th1[myc * PIPE_KMAX + lane] = t;
      }
      __builtin_amdgcn_s_setprio(0);
      st_keep = keep_row; st_th0 = th0_row; st_th1 = th1_eval; st_f1 = f1;
    }
    unsigned long long t_3 = dbg ? clk() : 0;
    lds_barrier();
    if (dbg) { unsigned long long t_4 = clk(); te += t_1 - t_0; tb1 += t_2 - t_1; to += t_3 - t_2; tb2 += t_4 - t_3; }
    // Row stores and bookkeeping of step v happen AFTER the barrier that releases the next evaluation: the owner
    // waves are the first of their SIMD to finish their MFMAs (~1400 ticks of slack), the stores ride in that slack
    // instead of sitting in the exposed owner phase.
    if (owner) {
      if (st_keep && v > burnin) {
        thin_ctr += 1;
        if (thin_ctr == thin) {
          thin_ctr = 0;
          if (plane) {
            *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sd_off + srow8)) = st_th0;
            if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sd_off + srow8)) = st_th1;
          }
          if (A.logpost && lane == 0 && !dbg) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = st_f1;
          srow8 += 8;
        }
      }
      if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
    }
  }
  if (dbg && lane == 0 && A.logpost) {   // stamps leave through the logpost buffer in this diagnostic mode
    double* d = A.logpost + (long long)A.nchains * A.S - 8 * ((long long)blockIdx.x * NW + wave + 1);
    d[0] = (double)te; d[1] = (double)tb1; d[2] = (double)to; d[3] = (double)tb2; d[4] = (double)nsteps; d[5] = (double)tf; d[6] = (double)tc; d[7] = (double)td;
  }
  if (owner) {
    if (plane) A.theta0[(long long)cl * k + lane] = th0;
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    }
  }
}

size_t mfma_lds_bytes() { return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 * MF_TCS); }

constexpr int SPEC_ALD = PIPE_KMAX + 1;                       // row stride of the k x k matrices in LDS
constexpr int SPEC_ADS = 7 * PIPE_KMAX + 2 * PIPE_KMAX * SPEC_ALD;  // doubles of adaptive state per chain

// Owner role of the specialised kernel for kernel_adapt (R/kernel_adapt.R:117-180) and kernel_ram
// (R/kernel_ram.R:123-158, unbounded parameters): same wave-collective arithmetic as mh_sweep_kernel (lanes = rows
// of Sigma / S, twin of the oracle's propose_adapt / propose_ram), state in LDS, variates from the HBM stream.
template <int KIND>
__device__ __forceinline__ void spec_owner_adaptive(const SweepArgs& A, int myc, int cl, double* s_th1, double* s_par,
                                                    unsigned* s_ready, unsigned* s_done, double* s_tr, double* ad) {
  const int lane = threadIdx.x & 63;
  const int k = A.k, kz = A.kz, nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  constexpr int LD = SPEC_ALD;
  double* th0 = ad;                    // [k]
  double* th1 = th0 + PIPE_KMAX;       // [k]
  double* vz = th1 + PIPE_KMAX;        // [kf] z / U of the pending proposal
  double* vv = vz + PIPE_KMAX;         // [kf] S U, or x
  double* vmp = vv + PIPE_KMAX;        // [kf] mean_prev
  double* vmt = vmp + PIPE_KMAX;       // [kf] mean_t
  double* vrs = vmt + PIPE_KMAX;       // [kf] running sum of ans rows
  double* SigA = vrs + PIPE_KMAX;      // [kf][LD]
  double* SigB = SigA + PIPE_KMAX * LD;
  __shared__ int s_which[4][PIPE_KMAX];
  int kf = 0;
  for (int j = 0; j < k; j++)
    if (!A.fixed[j]) { if (lane == 0) s_which[myc][kf] = j; kf++; }
  const int* which = s_which[myc];
  const double* s_mu = s_par, *s_lb = s_par + 2 * PIPE_KMAX, *s_ub = s_par + 3 * PIPE_KMAX;
  double f0 = 0.0;
  long long abs_iter = 0;
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0, have_mean = 0, nerr = 0;
  unsigned int srow8 = 0, bitword = 0;
  double* Scur = SigA;
  double* Salt = SigB;
  if (lane < k) { double t = A.theta0[(long long)cl * k + lane]; th0[lane] = t; th1[lane] = t; }
  for (int e = lane; e < kf * LD; e += 64) {
    const int a = e / LD, b = e % LD;
    SigA[e] = A.fresh ? ((a == b) ? 1.0 * A.eps : 0.0) : ((b < kf) ? A.Sigma[((long long)cl * kf + a) * kf + b] : 0.0);
    SigB[e] = 0.0;
  }
  if (!A.fresh) {
    abs_iter = A.abs_iter[cl];
    if (A.nerrors) nerr = A.nerrors[cl];
    if (KIND == FMCMC_KERNEL_ADAPT) {
      have_mean = A.have_mean[cl];
      if (lane < kf) vmp[lane] = A.mean_prev[(long long)cl * kf + lane];
    }
  }
  wave_sync_lds();
  const int jl = (lane < k) ? lane : 0;
  const unsigned int sd_off = (unsigned int)((((long long)cl * k + jl) * A.S) * 8);
  const unsigned int z_off = (unsigned int)((((long long)cl * nsteps) * kz + (lane < kz ? lane : 0)) * 8);
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.S) * 8);
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.fed_z) + (z_off + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  double z_nx = (lane < kz && nsteps >= 2) ? ld_z(1) : 0.0;
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;
  bool ram_gate = false;   // gate of the PENDING proposal (evaluated when it was made)
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };

  for (int v = 1; v <= nsteps; v++) {
    while (lds_ld_u32(&s_done[myc]) < 8u * (unsigned)v) __builtin_amdgcn_s_sleep(1);
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    const double tot = wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
    const double f1 = finish_logpost(A, th1, tot);
    bool keep_row = false, st_row = false;
    double st_th0 = 0.0, st_dr = 0.0;
    if (v == 1) {
      f0 = f1;
      if (lane < kf) vrs[lane] = th0[which[lane]];
      keep_row = true;
    } else if (status == FMCMC_CHAIN_OK) {
      const int i = v;
      if (KIND == FMCMC_KERNEL_RAM) {   // adaptation with f(theta1) of the pending (un-reflected) proposal :129-152
        if (ram_gate) {
          double a_n = fmh_exp(f1 - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          double eta = (double)kf * fmh_exp((-2.0 / 3.0) * fmh_log((double)i));
          if (eta > 1.0) eta = 1.0;
          double nrm2 = 0.0;
          for (int b = 0; b < kf; b++) nrm2 = fmh_fma(vz[b], vz[b], nrm2);
          double cp = (eta * (a_n - A.arate)) / nrm2;
          if (cp != 0.0 && fmh_isfinite(cp)) {
            const bool up = cp > 0.0;
            const double scl = fmh_sqrt(fmh_abs(cp));
            double w = (lane < kf) ? scl * vv[lane] : 0.0;
            bool fail = false;
            for (int j = 0; j < kf; j++) {
              double ljj = Scur[j * LD + j];
              double xj = shfl_d(w, j);
              double r2 = up ? fmh_fma(xj, xj, ljj * ljj) : fmh_fma(-xj, xj, ljj * ljj);
              if (!(r2 > 0.0) || !fmh_isfinite(r2)) { fail = true; break; }
              double r = fmh_sqrt(r2);
              double cc = r / ljj, ss = xj / ljj;
              if (lane == j) {
                Salt[j * LD + j] = r;
              } else if (lane > j && lane < kf) {
                double lij = Scur[lane * LD + j];
                double ln = (up ? fmh_fma(ss, w, lij) : fmh_fma(-ss, w, lij)) / cc;
                w = fmh_fma(-ss, ln, cc * w);
                Salt[lane * LD + j] = ln;
              }
            }
            wave_sync_lds();
            if (fail) nerr += 1;
            else { double* t = Scur; Scur = Salt; Salt = t; }
          }
        }
        abs_iter += 1;
      }
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (lane < k) A.status_theta[(long long)cl * k + lane] = th1[lane];
        flush_bits(i);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
        keep_row = true;
        // row i of ans / draws / logpost (the proposal th1 is still the evaluated one here)
        if (i > burnin) {
          thin_ctr += 1;
        }
        const bool acc = lu < ratio;
        const double dr = (lane < k) ? th1[lane] : 0.0;
        if (acc) {
          if (lane < k) th0[lane] = dr;
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        wave_sync_lds();
        if (i > burnin && thin_ctr == thin) {   // stored after the proposal is published (below)
          thin_ctr = 0;
          st_row = true;
          st_th0 = (lane < k) ? th0[lane] : 0.0;
          st_dr = dr;
        }
        if (KIND == FMCMC_KERNEL_ADAPT && lane < kf) vrs[lane] = vrs[lane] + th0[which[lane]];
        if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
      }
    }
    if (v == 1 && keep_row && 1 > burnin) {   // row 1 (R/mcmc.R:737-743)
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (lane < k) {
          *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sd_off + srow8)) = th0[lane];
          if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sd_off + srow8)) = th1[lane];
        }
        if (A.logpost && lane == 0) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = f1;
        srow8 += 8;
      }
    }
    // ---- proposal of loop step i = v + 1
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK) {
        const int i = v + 1;
        wave_sync_lds();
        if (lane < kz) vz[lane] = z_nx;
        z_nx = (lane < kz) ? ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1) : 0.0;
        wave_sync_lds();
        if (KIND == FMCMC_KERNEL_ADAPT) {
          if (A.until > (double)abs_iter && abs_iter > A.warmup && i > 2) {
            const double t = (double)(abs_iter - 1);
            double x = 0, mp = 0, mt = 0;
            if (lane < kf) {
              x = th0[which[lane]];
              mp = have_mean ? vmp[lane] : (vrs[lane] / (double)(i - 1));
              mt = (mp * t + x) / (t + 1);
              vv[lane] = x; vmp[lane] = mp; vmt[lane] = mt;
            }
            wave_sync_lds();
            if (lane < kf) {
              const double c1 = (t - 1) / t, c2 = 1.0 / t;
              for (int b = 0; b < kf; b++) {
                double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                double inner = t * (mp * vmp[b]) - (t + 1) * (mt * vmt[b]) + x * vv[b] + 1e-5 * ik;
                SigA[lane * LD + b] = c1 * SigA[lane * LD + b] + c2 * inner;
              }
            }
            wave_sync_lds();
            if (lane < kf) vmp[lane] = mt;
            have_mean = 1;
          }
          abs_iter += 1;
          bool notpd = false;
          for (int j = 0; j < kf; j++) {
            double sacc = 0.0;
            if (lane >= j && lane < kf) {
              sacc = SigA[lane * LD + j];
              for (int b = 0; b < j; b++) sacc = fmh_fma(-SigB[lane * LD + b], SigB[j * LD + b], sacc);
            }
            double d = shfl_d(sacc, j);
            if (!(d > 0.0) || !fmh_isfinite(d)) { notpd = true; break; }
            double ljj = fmh_sqrt(d);
            if (lane == j) SigB[j * LD + j] = ljj;
            else if (lane > j && lane < kf) SigB[lane * LD + j] = sacc / ljj;
            wave_sync_lds();
          }
          if (notpd) {
            status = FMCMC_CHAIN_NOT_PD;
            if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
            if (lane < k) A.status_theta[(long long)cl * k + lane] = th1[lane];
          } else {
            if (lane < k) th1[lane] = th0[lane];
            wave_sync_lds();
            if (lane < kf) {
              double sacc = 0.0;
              for (int b = 0; b <= lane; b++) sacc = fmh_fma(SigB[lane * LD + b], vz[b], sacc);
              const int j = which[lane];
              th1[j] = reflect1(th0[j] + (s_mu[j] + sacc), s_lb[j], s_ub[j]);
            }
          }
        } else {  // RAM P1 :123-126 (theta1 keeps its previous values in fixed coordinates)
          if (lane < kf) {
            double sacc = 0.0;
            for (int b = 0; b <= lane; b++) sacc = fmh_fma(Scur[lane * LD + b], vz[b], sacc);
            vv[lane] = sacc;
            const int j = which[lane];
            th1[j] = th0[j] + sacc;
          }
          ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && ((v + 1) % A.freq) == 0);
        }
        wave_sync_lds();
        if (lane < k) s_th1[myc * PIPE_KMAX + lane] = th1[lane];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_ready[myc], (unsigned)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (st_row) {   // row v of ans / draws / logpost, off the compute waves' critical path
      if (lane < k) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sd_off + srow8)) = st_th0;
        if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sd_off + srow8)) = st_dr;
      }
      if (A.logpost && lane == 0) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = f1;
      srow8 += 8;
    }
  }
  // ---- write state back
  wave_sync_lds();
  if (lane < k) A.theta0[(long long)cl * k + lane] = th0[lane];
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    A.abs_iter[cl] = abs_iter;
    if (A.nerrors) A.nerrors[cl] = nerr;
    if (KIND == FMCMC_KERNEL_ADAPT) A.have_mean[cl] = have_mean;
  }
  const double* Sfin = (KIND == FMCMC_KERNEL_RAM) ? Scur : SigA;
  for (int e = lane; e < kf * kf; e += 64) {
    const int a = e / kf, b = e % kf;
    A.Sigma[((long long)cl * kf + a) * kf + b] = Sfin[a * LD + b];
  }
  if (KIND == FMCMC_KERNEL_ADAPT && lane < kf) A.mean_prev[(long long)cl * kf + lane] = vmp[lane];
}

// Register-row variant of the adaptive owner for k <= SPEC_KA parameters, none fixed (C3, kernel_ram at k = 5):
// lane a keeps ROW a of Sigma / S (and of the Cholesky factor) in VGPRs and other rows' entries arrive by v_readlane
// (statically unrolled indices), so the Cholesky, the rank-1 update, the recursive covariance and the L z / S U
// products run without a single LDS round trip or ds_bpermute.  Same operations in the same order per matrix
// element as spec_owner_adaptive / the oracle, hence the same bits.
constexpr int SPEC_KA = 8;

template <int KIND>
__device__ __forceinline__ void spec_owner_adaptive_reg(const SweepArgs& A, int myc, int cl, double* s_th1, double* s_par,
                                                        unsigned* s_ready, unsigned* s_done, double* s_tr) {
  constexpr int KA = SPEC_KA;
  const int lane = threadIdx.x & 63;
  const int k = A.k, kf = A.k, kz = A.kz, nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const bool rl = lane < k;                 // row lane == parameter lane (no fixed parameters)
  const int jl = rl ? lane : 0;
  const double mu_l = A.mu[jl], lb_l = A.lb[jl], ub_l = A.ub[jl];
  double Srow[KA], Lrow[KA];                // Sigma (adapt) or S (ram) row `lane`; Cholesky factor row (adapt)
#pragma unroll
  for (int b = 0; b < KA; b++) {
    Lrow[b] = 0.0;
    Srow[b] = (rl && b < kf) ? (A.fresh ? ((b == lane) ? 1.0 * A.eps : 0.0) : A.Sigma[((long long)cl * kf + lane) * kf + b]) : 0.0;
  }
  double th0 = rl ? A.theta0[(long long)cl * k + lane] : 0.0, th1 = th0;
  double f0 = 0.0, mean_prev = 0.0, run_sum = 0.0, vv = 0.0, zcur = 0.0;
  long long abs_iter = 0;
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0, have_mean = 0, nerr = 0;
  unsigned int srow8 = 0, bitword = 0;
  if (!A.fresh) {
    abs_iter = A.abs_iter[cl];
    if (A.nerrors) nerr = A.nerrors[cl];
    if (KIND == FMCMC_KERNEL_ADAPT) {
      have_mean = A.have_mean[cl];
      if (rl) mean_prev = A.mean_prev[(long long)cl * kf + lane];
    }
  }
  const unsigned int sd_off = (unsigned int)((((long long)cl * k + jl) * A.S) * 8);
  const unsigned int z_off = (unsigned int)((((long long)cl * nsteps) * kz + jl) * 8);
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.S) * 8);
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.fed_z) + (z_off + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  double z_nx = (rl && nsteps >= 2) ? ld_z(1) : 0.0;
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;
  bool ram_gate = false;
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };
  auto logpost_of = [&](double tot, double sigma) -> double {   // Gaussian linreg closed form (same as the normal owners)
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
    while (lds_ld_u32(&s_done[myc]) < 8u * (unsigned)v) __builtin_amdgcn_s_sleep(1);
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    const double tot = wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
    const double f1 = logpost_of(tot, readlane_d(th1, k - 1));
    bool st_row = false;
    double st_th0 = 0.0;
    const double st_dr = th1;
    if (v == 1) {
      f0 = f1;
      run_sum = th0;
      if (1 > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
    } else if (status == FMCMC_CHAIN_OK) {
      const int i = v;
      if (KIND == FMCMC_KERNEL_RAM) {   // adaptation with f(theta1) of the pending proposal (R/kernel_ram.R:129-152)
        if (ram_gate) {
          double a_n = fmh_exp(f1 - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          double eta = (double)kf * fmh_exp((-2.0 / 3.0) * fmh_log((double)i));
          if (eta > 1.0) eta = 1.0;
          double nrm2 = 0.0;
#pragma unroll
          for (int b = 0; b < KA; b++)
            if (b < kf) { const double ub_ = readlane_d(zcur, b); nrm2 = fmh_fma(ub_, ub_, nrm2); }
          const double cp = (eta * (a_n - A.arate)) / nrm2;
          if (cp != 0.0 && fmh_isfinite(cp)) {
            const bool up = cp > 0.0;
            double w = rl ? fmh_sqrt(fmh_abs(cp)) * vv : 0.0;
            double Snew[KA];
#pragma unroll
            for (int b = 0; b < KA; b++) Snew[b] = Srow[b];
            bool fail = false;
#pragma unroll
            for (int j = 0; j < KA; j++) {
              if (j < kf && !fail) {
                const double ljj = readlane_d(Srow[j], j);
                const double xj = readlane_d(w, j);
                const double r2 = up ? fmh_fma(xj, xj, ljj * ljj) : fmh_fma(-xj, xj, ljj * ljj);
                if (!(r2 > 0.0) || !fmh_isfinite(r2)) {
                  fail = true;
                } else {
                  const double r = fmh_sqrt(r2);
                  const double cc = r / ljj, ss = xj / ljj;
                  if (lane == j) {
                    Snew[j] = r;
                  } else if (lane > j && rl) {
                    const double ln = (up ? fmh_fma(ss, w, Srow[j]) : fmh_fma(-ss, w, Srow[j])) / cc;
                    w = fmh_fma(-ss, ln, cc * w);
                    Snew[j] = ln;
                  }
                }
              }
            }
            if (fail) nerr += 1;
            else {
#pragma unroll
              for (int b = 0; b < KA; b++) Srow[b] = Snew[b];
            }
          }
        }
        abs_iter += 1;
      }
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
        if (lu < ratio) {
          th0 = th1;
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        if (i > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
        if (KIND == FMCMC_KERNEL_ADAPT) run_sum = run_sum + th0;
        if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
      }
    }
    // ---- proposal of loop step i = v + 1
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK) {
        const int i = v + 1;
        zcur = z_nx;
        z_nx = rl ? ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1) : 0.0;
        if (KIND == FMCMC_KERNEL_ADAPT) {
          if (A.until > (double)abs_iter && abs_iter > A.warmup && i > 2) {   // R/kernel_adapt.R:118-166
            const double t = (double)(abs_iter - 1);
            const double x = th0;
            const double mp = have_mean ? mean_prev : (run_sum / (double)(i - 1));
            const double mt = (mp * t + x) / (t + 1);
            const double c1 = (t - 1) / t, c2 = 1.0 / t;
#pragma unroll
            for (int b = 0; b < KA; b++) {
              if (b < kf) {
                const double mpb = readlane_d(mp, b), mtb = readlane_d(mt, b), xb = readlane_d(x, b);
                const double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                const double inner = t * (mp * mpb) - (t + 1) * (mt * mtb) + x * xb + 1e-5 * ik;
                Srow[b] = c1 * Srow[b] + c2 * inner;
              }
            }
            mean_prev = mt;
            have_mean = 1;
          }
          abs_iter += 1;
          // left-looking Cholesky: column j, lane = row (twin of oracle chol_lower_canon)
          bool notpd = false;
#pragma unroll
          for (int j = 0; j < KA; j++) {
            if (j < kf && !notpd) {
              double sacc = Srow[j];
#pragma unroll
              for (int b = 0; b < j; b++) sacc = fmh_fma(-Lrow[b], readlane_d(Lrow[b], j), sacc);
              const double d = readlane_d(sacc, j);
              if (!(d > 0.0) || !fmh_isfinite(d)) {
                notpd = true;
              } else {
                const double ljj = fmh_sqrt(d);
                if (lane == j) Lrow[j] = ljj;
                else if (lane > j) Lrow[j] = sacc / ljj;
              }
            }
          }
          if (notpd) {
            status = FMCMC_CHAIN_NOT_PD;
            if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
            if (rl) A.status_theta[(long long)cl * k + lane] = th1;
          } else {
            double sacc = 0.0;
#pragma unroll
            for (int b = 0; b < KA; b++)
              if (b < kf) { const double zb = readlane_d(zcur, b); if (b <= lane) sacc = fmh_fma(Lrow[b], zb, sacc); }
            th1 = reflect1(th0 + (mu_l + sacc), lb_l, ub_l);
          }
        } else {  // RAM P1 (R/kernel_ram.R:123-126)
          double sacc = 0.0;
#pragma unroll
          for (int b = 0; b < KA; b++)
            if (b < kf) { const double ub_ = readlane_d(zcur, b); if (b <= lane) sacc = fmh_fma(Srow[b], ub_, sacc); }
          vv = sacc;
          th1 = th0 + sacc;
          ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && ((v + 1) % A.freq) == 0);
        }
        if (rl) s_th1[myc * PIPE_KMAX + lane] = th1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_ready[myc], (unsigned)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (st_row) {   // row v of ans / draws / logpost, off the compute waves' critical path
      if (rl) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sd_off + srow8)) = st_th0;
        if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sd_off + srow8)) = st_dr;
      }
      if (A.logpost && lane == 0) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = f1;
      srow8 += 8;
    }
  }
  // ---- write state back
  if (rl) A.theta0[(long long)cl * k + lane] = th0;
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    A.abs_iter[cl] = abs_iter;
    if (A.nerrors) A.nerrors[cl] = nerr;
    if (KIND == FMCMC_KERNEL_ADAPT) A.have_mean[cl] = have_mean;
  }
#pragma unroll
  for (int b = 0; b < KA; b++)
    if (rl && b < kf) A.Sigma[((long long)cl * kf + lane) * kf + b] = (b <= lane || KIND == FMCMC_KERNEL_ADAPT) ? Srow[b] : 0.0;
  if (KIND == FMCMC_KERNEL_ADAPT && rl) A.mean_prev[(long long)cl * kf + lane] = mean_prev;
}

template <int P, int OPT, int KIND>
__global__ __launch_bounds__(SPEC_NT) void mh_sweep_spec(const SweepArgs A) {
  constexpr int CW = 4;
  static_assert(OPT % 2 == 0, "OPT must be even (y is read back in pairs)");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX] proposals read by the evaluation
  double* s_par = s_th1 + CW * PIPE_KMAX;          // [4][PIPE_KMAX]  mu, scale, lb, ub
  unsigned* s_ready = (unsigned*)(s_par + 4 * PIPE_KMAX);  // [CW] version of theta1[c] that is published
  unsigned* s_done = s_ready + CW;                         // [CW] partial arrivals (8 per version)
  double* s_tr = s_par + 4 * PIPE_KMAX + CW;       // [CW][8][PIPE_TRS] lane partials, transposed
  double* s_y = s_tr + CW * 8 * PIPE_TRS;          // [OPT/2][NT][2] this workgroup's copy of y
  double* s_ad = s_y + OPT * NT;                   // KIND >= 3: [CW][SPEC_ADS] adaptive per-chain state
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int ic = A.intercept;
  const bool dbg = (A.debug & 8) != 0;

  // ---- cooperative set-up (all 12 waves): y -> LDS, kernel constants, initial theta1, flags
  for (int e = tid; e < OPT * NT; e += SPEC_NT) {
    const int s = e / NT, t = e - s * NT;
    const long long i = (long long)t + (long long)NT * s;
    s_y[((s >> 1) * NT + t) * 2 + (s & 1)] = (i < A.n) ? A.y[i] : 0.0;
  }
  if (tid < k) {
    s_par[0 * PIPE_KMAX + tid] = A.mu[tid];
    s_par[1 * PIPE_KMAX + tid] = A.scale[tid];
    s_par[2 * PIPE_KMAX + tid] = A.lb[tid];
    s_par[3 * PIPE_KMAX + tid] = A.ub[tid];
  }
  if (tid < CW * PIPE_KMAX) {
    const int c = tid / PIPE_KMAX, j = tid - c * PIPE_KMAX;
    s_th1[tid] = (c < ncw && j < k) ? A.theta0[(cg0 + c) * k + j] : 0.0;
  }
  if (tid < CW) { s_ready[tid] = 1u; s_done[tid] = 0u; }
  __syncthreads();

  if (wave < SPEC_NCW) {
    // =========================== COMPUTE ROLE ===========================
    double xr[OPT][P > 0 ? P : 1];
    double wlast = 1.0;
#pragma unroll
    for (int s = 0; s < OPT; s++) {
      const long long i = (long long)tid + (long long)NT * s;
      const bool valid = i < A.n;
#pragma unroll
      for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
      if (s == OPT - 1) wlast = valid ? 1.0 : 0.0;
    }
    const int tr_slot = (tid & 7) * PIPE_TRS + (tid >> 3);
    const double2* yp = reinterpret_cast<const double2*>(s_y) + tid;
    unsigned long long tw = 0, te = 0;
    for (int v = 1; v <= nsteps; v++) {
      for (int c = 0; c < ncw; c++) {
        unsigned long long t_a = dbg ? clk() : 0;
        while (lds_ld_u32(&s_ready[c]) < (unsigned)v) __builtin_amdgcn_s_sleep(1);
        unsigned long long t_b = dbg ? clk() : 0;
        const double* t0 = s_th1 + c * PIPE_KMAX;
        const double m00 = ic ? t0[0] : 0.0;
        double b0[P > 0 ? P : 1];
#pragma unroll
        for (int j = 0; j < P; j++) b0[j] = t0[ic + j];
        double a0 = 0.0;
        // y pairs come from LDS three pairs (~30 FMAs) ahead of their use: LDS latency is ~130 cycles and only two
        // compute waves share the SIMD, so a one-pair lookahead leaves the FMA pipe waiting on lgkmcnt
        constexpr int YD = 3;
        double2 yq[YD];
#pragma unroll
        for (int d = 0; d < YD; d++) yq[d] = yp[(d < OPT / 2 ? d : OPT / 2 - 1) * NT];
#pragma unroll
        for (int s2 = 0; s2 < OPT / 2; s2++) {
          const double2 yy = yq[s2 % YD];
          if (s2 + YD < OPT / 2) yq[s2 % YD] = yp[(s2 + YD) * NT];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int s = 2 * s2 + h;
            const double yv = h ? yy.y : yy.x;
            double m0 = m00;
#pragma unroll
            for (int j = 0; j < P; j++) m0 = fmh_fma(xr[s][j], b0[j], m0);
            const double r0 = yv - m0;
            if (s == OPT - 1) a0 = fmh_fma(r0 * wlast, r0, a0);
            else a0 = fmh_fma(r0, r0, a0);
          }
        }
        s_tr[c * (8 * PIPE_TRS) + tr_slot] = a0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // partials landed before the arrival is visible
        if (lane == 0) __hip_atomic_fetch_add(&s_done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (dbg) { unsigned long long t_c = clk(); tw += t_b - t_a; te += t_c - t_b; }
      }
    }
    if (dbg && lane == 0 && A.draws) {
      double* d = A.draws + ((long long)blockIdx.x * 12 + wave) * 4;
      d[0] = (double)tw; d[1] = (double)te; d[2] = 0.0; d[3] = (double)nsteps;
    }
    return;
  }

  // =========================== OWNER ROLE ===========================
  const int myc = wave - SPEC_NCW;
  if (myc >= ncw) return;
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + myc);
  if constexpr (KIND == FMCMC_KERNEL_ADAPT || KIND == FMCMC_KERNEL_RAM) {
    bool nofixed = true;
    for (int j = 0; j < k; j++) nofixed = nofixed && (A.fixed[j] == 0);
    if (k <= SPEC_KA && nofixed && !(A.debug & 16))
      spec_owner_adaptive_reg<KIND>(A, myc, cl, s_th1, s_par, s_ready, s_done, s_tr);
    else
      spec_owner_adaptive<KIND>(A, myc, cl, s_th1, s_par, s_ready, s_done, s_tr, s_ad + myc * SPEC_ADS);
    return;
  }
  const bool plane = (lane < k);
  const int jl = plane ? lane : 0;
  const bool fixed_l = A.fixed[jl] != 0;
  int zidx = 0;
  for (int j = 0; j < jl; j++) zidx += A.fixed[j] ? 0 : 1;
  double th0 = plane ? A.theta0[(long long)cl * k + lane] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0;
  unsigned int srow8 = 0, bitword = 0;
  const unsigned int sd_off = (unsigned int)((((long long)cl * k + jl) * A.S) * 8);
  const unsigned int z_off = (unsigned int)((((long long)cl * nsteps) * kz + zidx) * 8);
  const unsigned int lp_off = (unsigned int)(((long long)cl * A.S) * 8);
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.fed_z) + (z_off + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  double z_nx = (plane && !fixed_l && nsteps >= 2) ? ld_z(1) : 0.0;   // variates of the NEXT proposal / decision
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;

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
    if (A.accept_bits && lane == 0)
      A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };

  unsigned long long tw = 0, tp = 0, tst = 0;
  for (int v = 1; v <= nsteps; v++) {
    // ---- wait for the 8 compute waves' partials of version v
    unsigned long long t_a = dbg ? clk() : 0;
    while (lds_ld_u32(&s_done[myc]) < 8u * (unsigned)v) __builtin_amdgcn_s_sleep(1);
    unsigned long long t_b = dbg ? clk() : 0;
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;  // this lane folds canonical lanes 8*lane .. 8*lane+7
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    const double tot = wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
    const double sigma = readlane_d(th1, k - 1);
    const double f1 = logpost_of(tot, sigma);
    const double th1_eval = th1;
    bool keep_row = false;
    if (v == 1) {                       // row 1: f0 = f(initial)
      f0 = uniform_d(f1);
      keep_row = true;
    } else if (status == FMCMC_CHAIN_OK) {
      const double ratio = f1 - f0;
      if (fmh_isnan(f1) || fmh_isnan(ratio)) {
        status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = v; }
        if (plane) A.status_theta[(long long)cl * k + lane] = th1;
        flush_bits(v);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];   // log u of step v + 1 (clamped), consumed a step later
        if (lu < ratio) {
          th0 = th1;
          f0 = uniform_d(f1);
          nacc += 1;
          bitword |= (1u << ((v - 1) & 31));
        }
        keep_row = true;
      }
    }
    const double th0_row = th0;
    // ---- proposal of step v + 1, published for the compute waves
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK && plane) {
        double t = th0;
        if (!fixed_l) {
          t = th0 + (s_par[0 * PIPE_KMAX + lane] + s_par[1 * PIPE_KMAX + lane] * z_nx);
          if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) t = reflect1(t, s_par[2 * PIPE_KMAX + lane], s_par[3 * PIPE_KMAX + lane]);
          z_nx = ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1);   // row of step v + 2 (clamped), awaited a step later
        }
        th1 = t;
        s_th1[myc * PIPE_KMAX + lane] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_ready[myc], (unsigned)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned long long t_c = dbg ? clk() : 0;
    // ---- stores last (off the compute waves' critical path)
    if (keep_row && v > burnin && !(dbg && A.draws)) {
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (plane) {
          *reinterpret_cast<double*>(reinterpret_cast<char*>(A.samples) + (sd_off + srow8)) = th0_row;
          if (A.draws) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.draws) + (sd_off + srow8)) = th1_eval;
        }
        if (A.logpost && lane == 0) *reinterpret_cast<double*>(reinterpret_cast<char*>(A.logpost) + (lp_off + srow8)) = f1;
        srow8 += 8;
      }
    }
    if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
    if (dbg) { unsigned long long t_d = clk(); tw += t_b - t_a; tp += t_c - t_b; tst += t_d - t_c; }
  }
  if (dbg && lane == 0 && A.draws) {
    double* d = A.draws + ((long long)blockIdx.x * 12 + wave) * 4;
    d[0] = (double)tw; d[1] = (double)tp; d[2] = (double)tst; d[3] = (double)nsteps;
  }
  // ---- write state back
  if (plane) A.theta0[(long long)cl * k + lane] = th0;
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
  }
}

size_t spec_lds_bytes(int opt, bool adaptive) {
  return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 + 4 * 8 * PIPE_TRS + (size_t)opt * NT + (adaptive ? 4 * SPEC_ADS : 0));
}

size_t pipe_lds_bytes(int opt) { return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 * 8 * PIPE_TRS + (size_t)opt * NT); }

// diagnostic: evaluates include/fmh_detmath.h / fmh_philox.h on the device (tests compare bitwise
// with the host build of the same headers)
__global__ void detmath_kernel(int which, const double* x, double* out, long long n,
                               unsigned long long seed) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r;
  switch (which) {
    case 0: r = fmh_log(v); break;
    case 1: r = fmh_exp(v); break;
    case 2: r = fmh_log1p(v); break;
    case 3: r = fmh_qnorm(v); break;
    case 4: r = fmh_log_accept_u(seed, (unsigned)(i & 0xffff), (unsigned)(i >> 16)); break;
    case 5: r = fmh_normal(seed, (unsigned)(i & 0xffff), (unsigned)(i >> 16), (unsigned)(i % 7)); break;
    case 6: r = fmh_student_t(seed, (unsigned)(i & 0xffff), (unsigned)(i >> 16), (unsigned)(i % 7), v); break;
    case 7: r = fmh_sqrt(v); break;
    case 8: r = 1.0 / v; break;
    case 9: r = fmh_log1p_exp_nonpos(v); break;   // the fused softplus tail of the logistic family
    case 10: r = fmh_unif(seed, (unsigned)(i & 0xffff), (unsigned)(i >> 16), (unsigned)(i % 7)); break;
    case 12: r = fmh_tan_0_halfpi(v); break;
    default: r = fmh_nan();
  }
  out[i] = r;
}

size_t sweep_lds_bytes(int k, int kf, int kind, int CW, int tb, int kz, bool resident) {
  size_t d = 4 * (size_t)k + (k / 2 + 1) + (size_t)NW * CW + 1 + (size_t)CW * tb * (kz + 1) +
             (resident ? (size_t)CW * NT : 0) + (size_t)CW * chain_lds_doubles(k, kf, kind);
  return d * sizeof(double);
}

}  // namespace

// ==============================================================================================
// C-ABI
// ==============================================================================================
extern "C" {

int fmcmc_abi_version(void) { return FMCMC_ABI_VERSION; }
const char* fmcmc_last_error(void) { return g_err; }

int fmcmc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int64_t fmcmc_kept_rows(int64_t nsteps, int64_t burnin, int64_t thin) {
  if (thin < 1 || burnin >= nsteps) return 0;
  return (nsteps - burnin) / thin;
}

static int count_free(const fmcmc_kernel* kn, const uint8_t* fixed_host) {
  int kf = 0;
  for (int j = 0; j < kn->k; j++)
    if (!fixed_host[j]) kf++;
  return kf;
}

// Argument checks with the reference's own messages (R/mcmc.R:501-520, R/kernel.R:9,129-132,
// R/kernel_normal.R:134-135). Pointers inside `kernel` must be HOST pointers here.
static bool is_simple_kind(int kind) {
  return kind == FMCMC_KERNEL_NORMAL || kind == FMCMC_KERNEL_NORMAL_REFLECTIVE || kind == FMCMC_KERNEL_UNIF ||
         kind == FMCMC_KERNEL_UNIF_REFLECTIVE || kind == FMCMC_KERNEL_NMIRROR || kind == FMCMC_KERNEL_UMIRROR;
}
static int variates_per_step(const fmcmc_kernel* kn, int kf) {  // single-parameter schemes draw one variate per step
  return (is_simple_kind(kn->kind) && kn->scheme != FMCMC_SCHEME_JOINT) ? 1 : kf;
}

int fmcmc_validate(const fmcmc_model* m, const fmcmc_kernel* kn, const fmcmc_run* run) {
  if (!m || !kn || !run) { set_err("null argument"); return FMCMC_ERR_ARG; }
  if (run->nchains < 1) { set_err("`nchains` must be an integer greater than 1."); return FMCMC_ERR_ARG; }
  if (run->burnin >= run->nsteps) {
    set_err("-burnin- (%lld) cannot be >= than -nsteps- (%lld).", (long long)run->burnin, (long long)run->nsteps);
    return FMCMC_ERR_ARG;
  }
  if (run->thin >= run->nsteps) {
    set_err("-thin- (%lld) cannot be > than -nsteps- (%lld).", (long long)run->thin, (long long)run->nsteps);
    return FMCMC_ERR_ARG;
  }
  if (run->thin < 1) { set_err("-thin- should be >= 1."); return FMCMC_ERR_ARG; }
  if (kn->k < 1 || kn->k > FMCMC_MAX_K) {
    set_err("number of parameters k=%d outside [1, %d]", kn->k, FMCMC_MAX_K);
    return FMCMC_ERR_UNSUPPORTED;
  }
  int kexp = -1;
  switch (m->family) {
    case FMCMC_FAM_GAUSSIAN_LINREG: kexp = (m->intercept ? 1 : 0) + m->p + 1; break;
    case FMCMC_FAM_LOGISTIC: kexp = (m->intercept ? 1 : 0) + m->p; break;
    case FMCMC_FAM_IID_NORMAL: kexp = 2; break;
    default: set_err("unknown log-posterior family %d", m->family); return FMCMC_ERR_ARG;
  }
  if (kexp != kn->k) {
    set_err("Incorrect length of -initial-: the model has %d parameters, the kernel %d.", kexp, kn->k);
    return FMCMC_ERR_ARG;
  }
  if (m->n < 1) { set_err("the model needs at least one observation"); return FMCMC_ERR_ARG; }
  if (kn->kind < FMCMC_KERNEL_NORMAL || kn->kind > FMCMC_KERNEL_UMIRROR) {
    set_err("unknown kernel kind %d", kn->kind);
    return FMCMC_ERR_ARG;
  }
  const bool simple = is_simple_kind(kn->kind);
  if (simple && (kn->scheme < FMCMC_SCHEME_JOINT || kn->scheme > FMCMC_SCHEME_EXPLICIT)) {
    set_err("-scheme- update must be either an integer sequence, 'joint', 'ordered', or 'random'.");
    return FMCMC_ERR_ARG;
  }
  if (kn->fixed && kn->lb && kn->ub) {
    int kf = count_free(kn, kn->fixed);
    if (kf == 0) {
      set_err("The number of parameters to update, i.e. not fixed, cannot be zero. "
              "Check the value -fixed- in the kernel initialization.");
      return FMCMC_ERR_ARG;
    }
    if (kn->kind != FMCMC_KERNEL_NORMAL && kn->kind != FMCMC_KERNEL_UNIF)
      for (int j = 0; j < kn->k; j++)
        if (!(kn->ub[j] > kn->lb[j])) { set_err("-ub- cannot be <= than -lb-."); return FMCMC_ERR_ARG; }
    if ((kn->kind == FMCMC_KERNEL_UNIF || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE) && kn->scale)
      for (int j = 0; j < kn->k; j++)   // scale = max. - min. (R/kernel_unif.R:55-56, :123-124)
        if (!(kn->scale[j] > 0.0)) { set_err("-max.- cannot be <= than -min.-."); return FMCMC_ERR_ARG; }
    if (simple && kn->scheme == FMCMC_SCHEME_EXPLICIT) {  // R/kernel.R:72-90
      if (!kn->scheme_seq || kn->scheme_len != kf) {
        set_err("When setting the update scheme, it should have the same length as the number of variables that will "
                "not be fixed. Right now length(scheme) = %d while sum(!fixed) = %d.", kn->scheme_seq ? kn->scheme_len : 0, kf);
        return FMCMC_ERR_ARG;
      }
      for (int j = 0; j < kn->k; j++) {
        if (kn->fixed[j]) continue;
        bool found = false;
        for (int a = 0; a < kn->scheme_len; a++) found = found || (kn->scheme_seq[a] == j);
        if (!found) {
          set_err("One or more variables was not included in the ordering sequence. Only variables that are not fixed "
                  "can be included in this list.");
          return FMCMC_ERR_ARG;
        }
      }
    }
  }
  if (kn->kind == FMCMC_KERNEL_ADAPT && (kn->freq < 1 || kn->bw < 0)) {
    set_err("-freq- must be >= 1 and -bw- >= 0 (got freq=%d, bw=%d)", kn->freq, kn->bw);
    return FMCMC_ERR_ARG;
  }
  if (kn->kind == FMCMC_KERNEL_RAM && kn->freq < 1) { set_err("-freq- must be >= 1."); return FMCMC_ERR_ARG; }
  if (kn->kind == FMCMC_KERNEL_ADAPT && kn->bw > 0 && kn->bw > kn->warmup) {
    set_err("The `warmup` parameter must be greater than `bw`.");
    return FMCMC_ERR_ARG;
  }
  if (run->rng_mode == FMCMC_RNG_FED && (!run->fed_logu || !run->fed_z)) {
    set_err("rng_mode = FED needs fed_logu and fed_z");
    return FMCMC_ERR_ARG;
  }
  return FMCMC_OK;
}

// kernel->fixed etc. are DEVICE pointers here; kf and bounds info come via `kf`/`ram_bounded`.
static int launch_sweep(const fmcmc_model* m, const fmcmc_kernel* kn_in, const fmcmc_run* run,
                        fmcmc_state* st, fmcmc_out* out, int kf, int ram_bounded, hipStream_t stream) {
  SweepArgs A;
  memset(&A, 0, sizeof(A));
  // the uniform kernels ARE the normal kernels with mu = min., scale = max. - min. and U(0,1) variates
  fmcmc_kernel ke = *kn_in;
  if (ke.kind == FMCMC_KERNEL_UNIF) { ke.kind = FMCMC_KERNEL_NORMAL; A.variate = 1; }
  if (ke.kind == FMCMC_KERNEL_UNIF_REFLECTIVE) { ke.kind = FMCMC_KERNEL_NORMAL_REFLECTIVE; A.variate = 1; }
  if (ke.kind == FMCMC_KERNEL_UMIRROR) A.variate = 1;
  const bool mirror = (ke.kind == FMCMC_KERNEL_NMIRROR || ke.kind == FMCMC_KERNEL_UMIRROR);
  if (mirror && (!st->mirror_mu || !st->mirror_scale || !st->obs_arate || !st->abs_iter)) {
    set_err("mirror kernels need state->mirror_mu, mirror_scale, obs_arate and abs_iter");
    return FMCMC_ERR_ARG;
  }
  A.nadapt = kn_in->nadapt; A.mirror_mu = st->mirror_mu; A.mirror_scale = st->mirror_scale; A.obs_arate = st->obs_arate;
  const fmcmc_kernel* kn = &ke;
  if ((kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE || mirror) && kn->scheme == FMCMC_SCHEME_RANDOM && run->rng_mode == FMCMC_RNG_FED &&
      !st->scheme_cols) {
    set_err("rng_mode = FED with scheme = 'random' needs state->scheme_cols");
    return FMCMC_ERR_ARG;
  }
  A.bw = (kn->kind == FMCMC_KERNEL_ADAPT) ? kn->bw : 0; A.Sd = kn->Sd;
  const bool adapt_hist = (kn->kind == FMCMC_KERNEL_ADAPT && (kn->bw > 0 || kn->freq > 1));
  if (adapt_hist) {   // ring of the last rows of every chain (the reference reads them from env$ans)
    A.hist_rows = (kn->bw - 1 > kn->freq) ? kn->bw - 1 : kn->freq;
    hipError_t eh = hipMallocAsync((void**)&A.hist, sizeof(double) * (size_t)run->nchains * (size_t)A.hist_rows * (size_t)kf, stream);
    if (eh != hipSuccess) { set_err("hipMallocAsync(adapt history) failed: %s", hipGetErrorString(eh)); return FMCMC_ERR_DEVICE; }
  }
  A.freq = kn->freq < 1 ? 1 : kn->freq; A.scheme_seq = kn->scheme_seq; A.scheme_len = kn->scheme_len;
  A.constr = (kn->kind == FMCMC_KERNEL_RAM) ? kn->constr : nullptr; A.scheme_cols = st->scheme_cols;
  A.family = m->family; A.p = m->p; A.intercept = m->intercept ? 1 : 0; A.guard = m->guard ? 1 : 0;
  A.n = m->n; A.X = m->X; A.y = m->y; A.prior_div = m->prior_div;
  A.kind = kn->kind; A.k = kn->k; A.scheme = kn->scheme; A.warmup = kn->warmup;
  A.until = kn->until; A.eps = kn->eps; A.arate = kn->arate;
  A.mu = kn->mu; A.scale = kn->scale; A.lb = kn->lb; A.ub = kn->ub; A.fixed = kn->fixed;
  A.nchains = run->nchains; A.nsteps = run->nsteps; A.burnin = run->burnin; A.thin = run->thin;
  A.S = fmcmc_kept_rows(run->nsteps, run->burnin, run->thin);
  A.chain_base = run->chain_base; A.step_base = run->step_base; A.seed = run->seed;
  A.rng_mode = run->rng_mode; A.fresh = st->fresh; A.ram_bounded = ram_bounded;
  A.kz = variates_per_step(kn, kf);
  A.fed_logu = run->fed_logu; A.fed_z = run->fed_z;
  { const char* dbg = getenv("FMCMC_AMD_DEBUG_MODE"); A.debug = dbg ? atoi(dbg) : 0; }  // timing ablations only
  A.theta0 = st->theta0; A.f0 = st->f0; A.abs_iter = (long long*)st->abs_iter; A.Sigma = st->Sigma;
  A.mean_prev = st->mean_prev; A.have_mean = st->have_mean; A.nerrors = st->nerrors;
  A.samples = out->samples; A.logpost = out->logpost; A.draws = out->draws;
  A.accept_count = (long long*)out->accept_count; A.accept_bits = out->accept_bits;
  A.status = out->status; A.status_step = (long long*)out->status_step; A.status_theta = out->status_theta;

  // ---- launch geometry
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  // register-resident variant: Gaussian linreg whose data fits the VGPR budget of 512 threads
  int res_p = -1, res_opt = 0;
  const char* force = getenv("FMCMC_AMD_FORCE_STREAMED");
  if (!(force && force[0] == '1') && m->family == FMCMC_FAM_GAUSSIAN_LINREG && !mirror) {
    static const int variants[][2] = {{1, 4}, {3, 20}};
    for (auto& v : variants)
      if (m->p == v[0] && m->n > (long long)NT * (v[1] - RES_MASKED) && m->n <= (long long)NT * v[1]) {
        res_p = v[0];
        res_opt = v[1];
      }
  }
  const bool resident = res_p >= 0;
  // chains per workgroup: fill the CUs first, then stack chains on a workgroup
  int cw = 1;
  if (resident) {
    cw = 4;
  } else {
    while (cw < NW && (long long)cw * ncu < run->nchains) cw <<= 1;
    const char* cwenv = getenv("FMCMC_AMD_CW");   // diagnosis: chains per workgroup of the streamed kernel (1, 2, 4, 8)
    if (cwenv && (cwenv[0] == '1' || cwenv[0] == '2' || cwenv[0] == '4' || cwenv[0] == '8')) cw = cwenv[0] - '0';
  }
  int tb = 32;
  while (tb > 1 && sweep_lds_bytes(kn->k, kf, kn->kind, cw, tb, A.kz, resident) > 60 * 1024) tb >>= 1;
  while (cw > 1 && !resident && sweep_lds_bytes(kn->k, kf, kn->kind, cw, tb, A.kz, resident) > 150 * 1024) cw >>= 1;
  A.tb = tb;
  size_t lds = sweep_lds_bytes(kn->k, kf, kn->kind, cw, tb, A.kz, resident);
  if (lds > 160 * 1024) { set_err("LDS budget exceeded (k=%d)", kn->k); return FMCMC_ERR_UNSUPPORTED; }
  const long long nblk = (run->nchains + cw - 1) / cw;
  hipError_t e = hipSuccess;
#define LAUNCH(CWV, PV, OV, KV)                                                                      \
  do {                                                                                               \
    if (lds > 48 * 1024)                                                                             \
      e = hipFuncSetAttribute((const void*)mh_sweep_kernel<CWV, PV, OV, KV>,                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    if (e == hipSuccess)                                                                             \
      hipLaunchKernelGGL((mh_sweep_kernel<CWV, PV, OV, KV>), dim3((unsigned)nblk), dim3(NT), lds, stream, A); \
  } while (0)
#define LAUNCH_KIND(CWV, PV, OV)                                                                     \
  switch (kn->kind) {                                                                                \
    case FMCMC_KERNEL_NORMAL: LAUNCH(CWV, PV, OV, 1); break;                                         \
    case FMCMC_KERNEL_NORMAL_REFLECTIVE: LAUNCH(CWV, PV, OV, 2); break;                              \
    case FMCMC_KERNEL_ADAPT: LAUNCH(CWV, PV, OV, 3); break;                                          \
    default: LAUNCH(CWV, PV, OV, 4); break;                                                          \
  }
  // software-pipelined fast path: normal kernels, joint scheme, k <= 16, linreg data in registers
  const char* nopipe = getenv("FMCMC_AMD_NO_PIPE");
  const char* nospec0 = getenv("FMCMC_AMD_NO_SPEC");
  int pipe_opt = 0, mfma_ng = 0;
  if (!(force && force[0] == '1') && !(nopipe && nopipe[0] == '1') && m->family == FMCMC_FAM_GAUSSIAN_LINREG &&
      (kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE ||
       (((kn->kind == FMCMC_KERNEL_ADAPT && !adapt_hist) || (kn->kind == FMCMC_KERNEL_RAM && !ram_bounded && !kn->constr)) && !(nospec0 && nospec0[0] == '1'))) &&
      (kn->scheme == FMCMC_SCHEME_JOINT || kn->kind >= FMCMC_KERNEL_ADAPT) && kn->k <= PIPE_KMAX &&
      (unsigned long long)run->nchains * kn->k * (unsigned long long)A.S * 8ull < (1ull << 32) &&
      (unsigned long long)run->nchains * (unsigned long long)run->nsteps * (unsigned long long)A.kz * 8ull < (1ull << 32)) {
    if (m->p == 3 && m->n > (long long)NT * 19 && m->n <= (long long)NT * 20) pipe_opt = 20;
    if (m->p == 1 && m->n > (long long)NT * 1 && m->n <= (long long)NT * 2) pipe_opt = 2;
    // fp64-MFMA evaluation: general in n and p up to what 80 operand registers per lane hold (normal / uniform kernels)
    const char* usemf0 = getenv("FMCMC_AMD_MFMA");
    if (!(usemf0 && usemf0[0] == '0') && kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && kn->scheme == FMCMC_SCHEME_JOINT) {
      if (m->p <= 3 && m->n <= (long long)NT * 20) mfma_ng = 1;
      else if (m->p <= 7 && m->n <= (long long)NT * 10) mfma_ng = 2;
      // the wave-specialised VALU kernel overlaps owners and evaluation; at its small shape that beats the MFMAs
      if (pipe_opt == 2 && !(nospec0 && nospec0[0] == '1')) mfma_ng = 0;
    }
  }
  if (pipe_opt || mfma_ng) {
    const size_t plds = pipe_opt ? pipe_lds_bytes(pipe_opt) : 0;
    const long long pblk = (run->nchains + 3) / 4;
    double* ws = nullptr;
    if (A.rng_mode == FMCMC_RNG_PHILOX) {
      // materialise the canonical stream: [C][nsteps] log u, then [C][nsteps][kz] z
      const size_t items = (size_t)run->nchains * (size_t)run->nsteps;
      e = hipMallocAsync((void**)&ws, sizeof(double) * items * (size_t)(A.kz + 1), stream);
      if (e != hipSuccess) { set_err("hipMallocAsync(rng stream) failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
      hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                         (unsigned long long)run->seed, (long long)run->step_base, (long long)run->chain_base,
                         (long long)run->nchains, (long long)run->nsteps, A.kz,
                         (kn->kind == FMCMC_KERNEL_RAM) ? kf : (A.variate == 1 ? -1 : 0), ws, ws + items);
      A.fed_logu = ws;
      A.fed_z = ws + items;
      A.rng_mode = FMCMC_RNG_FED;
    }
#define LAUNCH_PIPE(PV, OV, KV)                                                                        \
    do {                                                                                               \
      if (plds > 48 * 1024)                                                                            \
        e = hipFuncSetAttribute((const void*)mh_sweep_pipe<PV, OV, KV>,                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);                \
      if (e == hipSuccess)                                                                             \
        hipLaunchKernelGGL((mh_sweep_pipe<PV, OV, KV>), dim3((unsigned)pblk), dim3(NT), plds, stream, A); \
    } while (0)
    const char* nospec = getenv("FMCMC_AMD_NO_SPEC");
    if (mfma_ng) {
      const size_t mlds = mfma_lds_bytes();
      const int ns = (int)((m->n + NT - 1) / NT);   // observation slots of 512
#define LAUNCH_MFMA(KV, GV, SV) hipLaunchKernelGGL((mh_sweep_mfma<KV, GV, SV, false>), dim3((unsigned)pblk), dim3(NT), mlds, stream, A)
      if ((A.debug & 8) && mfma_ng == 1 && ns == 20) {
        if (kn->kind == FMCMC_KERNEL_NORMAL) hipLaunchKernelGGL((mh_sweep_mfma<1, 1, 20, true>), dim3((unsigned)pblk), dim3(NT), mlds, stream, A);
        else hipLaunchKernelGGL((mh_sweep_mfma<2, 1, 20, true>), dim3((unsigned)pblk), dim3(NT), mlds, stream, A);
      } else if (mfma_ng == 2 && kn->kind == FMCMC_KERNEL_NORMAL) {
        switch (ns) {
          case 1: LAUNCH_MFMA(1, 2, 1); break;
          case 2: LAUNCH_MFMA(1, 2, 2); break;
          case 3: LAUNCH_MFMA(1, 2, 3); break;
          case 4: LAUNCH_MFMA(1, 2, 4); break;
          case 5: LAUNCH_MFMA(1, 2, 5); break;
          case 6: LAUNCH_MFMA(1, 2, 6); break;
          case 7: LAUNCH_MFMA(1, 2, 7); break;
          case 8: LAUNCH_MFMA(1, 2, 8); break;
          case 9: LAUNCH_MFMA(1, 2, 9); break;
          case 10: LAUNCH_MFMA(1, 2, 10); break;
          default: break;
        }
      } else if (mfma_ng == 2) {
        switch (ns) {
          case 1: LAUNCH_MFMA(2, 2, 1); break;
          case 2: LAUNCH_MFMA(2, 2, 2); break;
          case 3: LAUNCH_MFMA(2, 2, 3); break;
          case 4: LAUNCH_MFMA(2, 2, 4); break;
          case 5: LAUNCH_MFMA(2, 2, 5); break;
          case 6: LAUNCH_MFMA(2, 2, 6); break;
          case 7: LAUNCH_MFMA(2, 2, 7); break;
          case 8: LAUNCH_MFMA(2, 2, 8); break;
          case 9: LAUNCH_MFMA(2, 2, 9); break;
          case 10: LAUNCH_MFMA(2, 2, 10); break;
          default: break;
        }
      } else if (kn->kind == FMCMC_KERNEL_NORMAL) {
        switch (ns) {
          case 1: LAUNCH_MFMA(1, 1, 1); break;
          case 2: LAUNCH_MFMA(1, 1, 2); break;
          case 3: LAUNCH_MFMA(1, 1, 3); break;
          case 4: LAUNCH_MFMA(1, 1, 4); break;
          case 5: LAUNCH_MFMA(1, 1, 5); break;
          case 6: LAUNCH_MFMA(1, 1, 6); break;
          case 7: LAUNCH_MFMA(1, 1, 7); break;
          case 8: LAUNCH_MFMA(1, 1, 8); break;
          case 9: LAUNCH_MFMA(1, 1, 9); break;
          case 10: LAUNCH_MFMA(1, 1, 10); break;
          case 11: LAUNCH_MFMA(1, 1, 11); break;
          case 12: LAUNCH_MFMA(1, 1, 12); break;
          case 13: LAUNCH_MFMA(1, 1, 13); break;
          case 14: LAUNCH_MFMA(1, 1, 14); break;
          case 15: LAUNCH_MFMA(1, 1, 15); break;
          case 16: LAUNCH_MFMA(1, 1, 16); break;
          case 17: LAUNCH_MFMA(1, 1, 17); break;
          case 18: LAUNCH_MFMA(1, 1, 18); break;
          case 19: LAUNCH_MFMA(1, 1, 19); break;
          case 20: LAUNCH_MFMA(1, 1, 20); break;
          default: break;
        }
      } else {
        switch (ns) {
          case 1: LAUNCH_MFMA(2, 1, 1); break;
          case 2: LAUNCH_MFMA(2, 1, 2); break;
          case 3: LAUNCH_MFMA(2, 1, 3); break;
          case 4: LAUNCH_MFMA(2, 1, 4); break;
          case 5: LAUNCH_MFMA(2, 1, 5); break;
          case 6: LAUNCH_MFMA(2, 1, 6); break;
          case 7: LAUNCH_MFMA(2, 1, 7); break;
          case 8: LAUNCH_MFMA(2, 1, 8); break;
          case 9: LAUNCH_MFMA(2, 1, 9); break;
          case 10: LAUNCH_MFMA(2, 1, 10); break;
          case 11: LAUNCH_MFMA(2, 1, 11); break;
          case 12: LAUNCH_MFMA(2, 1, 12); break;
          case 13: LAUNCH_MFMA(2, 1, 13); break;
          case 14: LAUNCH_MFMA(2, 1, 14); break;
          case 15: LAUNCH_MFMA(2, 1, 15); break;
          case 16: LAUNCH_MFMA(2, 1, 16); break;
          case 17: LAUNCH_MFMA(2, 1, 17); break;
          case 18: LAUNCH_MFMA(2, 1, 18); break;
          case 19: LAUNCH_MFMA(2, 1, 19); break;
          case 20: LAUNCH_MFMA(2, 1, 20); break;
          default: break;
        }
      }
#undef LAUNCH_MFMA
    } else
    if (!(nospec && nospec[0] == '1')) {
      const size_t slds = spec_lds_bytes(pipe_opt, kn->kind >= FMCMC_KERNEL_ADAPT);
#define LAUNCH_SPEC(PV, OV, KV)                                                                        \
      do {                                                                                             \
        if (slds > 48 * 1024)                                                                          \
          e = hipFuncSetAttribute((const void*)mh_sweep_spec<PV, OV, KV>,                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);              \
        if (e == hipSuccess)                                                                           \
          hipLaunchKernelGGL((mh_sweep_spec<PV, OV, KV>), dim3((unsigned)pblk), dim3(SPEC_NT), slds, stream, A); \
      } while (0)
      if (pipe_opt == 20) {
        switch (kn->kind) {
          case FMCMC_KERNEL_NORMAL: LAUNCH_SPEC(3, 20, 1); break;
          case FMCMC_KERNEL_NORMAL_REFLECTIVE: LAUNCH_SPEC(3, 20, 2); break;
          case FMCMC_KERNEL_ADAPT: LAUNCH_SPEC(3, 20, 3); break;
          default: LAUNCH_SPEC(3, 20, 4); break;
        }
      } else {
        switch (kn->kind) {
          case FMCMC_KERNEL_NORMAL: LAUNCH_SPEC(1, 2, 1); break;
          case FMCMC_KERNEL_NORMAL_REFLECTIVE: LAUNCH_SPEC(1, 2, 2); break;
          case FMCMC_KERNEL_ADAPT: LAUNCH_SPEC(1, 2, 3); break;
          default: LAUNCH_SPEC(1, 2, 4); break;
        }
      }
#undef LAUNCH_SPEC
    } else
    if (pipe_opt == 20 && kn->kind == FMCMC_KERNEL_NORMAL) LAUNCH_PIPE(3, 20, 1);
    else if (pipe_opt == 20) LAUNCH_PIPE(3, 20, 2);
    else if (kn->kind == FMCMC_KERNEL_NORMAL) LAUNCH_PIPE(1, 2, 1);
    else LAUNCH_PIPE(1, 2, 2);
#undef LAUNCH_PIPE
    if (ws) (void)hipFreeAsync(ws, stream);
  } else
  if (resident && res_p == 1) { LAUNCH_KIND(4, 1, 4); }
  else if (resident && res_p == 3) { LAUNCH_KIND(4, 3, 20); }
  else switch (cw) {
    case 1: LAUNCH(1, -1, 0, 0); break;
    case 2: LAUNCH(2, -1, 0, 0); break;
    case 4: LAUNCH(4, -1, 0, 0); break;
    default: LAUNCH(8, -1, 0, 0); break;
  }
#undef LAUNCH_KIND
#undef LAUNCH
  if (A.hist) (void)hipFreeAsync(A.hist, stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { set_err("HIP launch failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
  return FMCMC_OK;
}

int fmcmc_rng_stream_dev(uint64_t seed, int64_t step_base, int64_t chain_base, int64_t nchains, int64_t nsteps,
                         int32_t kz, int32_t student_df, double* logu, double* z, void* hip_stream) {
  if (!logu || !z || nchains < 1 || nsteps < 1 || kz < 1) { set_err("fmcmc_rng_stream_dev: bad argument"); return FMCMC_ERR_ARG; }
  const size_t items = (size_t)nchains * (size_t)nsteps;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     (unsigned long long)seed, (long long)step_base, (long long)chain_base, (long long)nchains,
                     (long long)nsteps, (int)kz, (int)student_df, logu, z);
  return hipGetLastError() == hipSuccess ? FMCMC_OK : FMCMC_ERR_DEVICE;
}

int fmcmc_detmath_dev(int which, const double* x, double* out, int64_t n, uint64_t seed, void* hip_stream) {
  if (n <= 0) return FMCMC_OK;
  hipLaunchKernelGGL(detmath_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     which, x, out, (long long)n, (unsigned long long)seed);
  return hipGetLastError() == hipSuccess ? FMCMC_OK : FMCMC_ERR_DEVICE;
}

int fmcmc_mcmc_run_dev(const fmcmc_model* m, const fmcmc_kernel* kn, const fmcmc_run* run,
                       fmcmc_state* st, fmcmc_out* out, void* hip_stream) {
  if (!m || !kn || !run || !st || !out) { set_err("null argument"); return FMCMC_ERR_ARG; }
  // `fixed`, `lb`, `ub` live on the device: fetch the few bytes the launch geometry needs.
  uint8_t fx[MAXK];
  double lb[MAXK], ub[MAXK];
  if (kn->k < 1 || kn->k > MAXK) { set_err("k=%d outside [1,%d]", kn->k, MAXK); return FMCMC_ERR_UNSUPPORTED; }
  hipStream_t stream = (hipStream_t)hip_stream;
  if (hipMemcpyAsync(fx, kn->fixed, kn->k, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipMemcpyAsync(lb, kn->lb, kn->k * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipMemcpyAsync(ub, kn->ub, kn->k * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) {
    set_err("cannot read kernel parameters from device memory");
    return FMCMC_ERR_DEVICE;
  }
  fmcmc_kernel kh = *kn;
  kh.fixed = fx; kh.lb = lb; kh.ub = ub;
  double sc[MAXK];
  int32_t seq[MAXK];
  const bool unif = (kn->kind == FMCMC_KERNEL_UNIF || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE);
  const bool expl = (is_simple_kind(kn->kind) && kn->scheme == FMCMC_SCHEME_EXPLICIT && kn->scheme_seq &&
                     kn->scheme_len >= 1 && kn->scheme_len <= MAXK);
  if ((unif && hipMemcpyAsync(sc, kn->scale, kn->k * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) ||
      (expl && hipMemcpyAsync(seq, kn->scheme_seq, kn->scheme_len * sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) ||
      hipStreamSynchronize(stream) != hipSuccess) {
    set_err("cannot read kernel parameters from device memory");
    return FMCMC_ERR_DEVICE;
  }
  kh.scale = unif ? sc : nullptr;
  kh.scheme_seq = expl ? seq : nullptr;
  int rc = fmcmc_validate(m, &kh, run);
  if (rc != FMCMC_OK) return rc;
  int kf = count_free(kn, fx);
  int bounded = 0;
  for (int j = 0; j < kn->k; j++)
    if (!fx[j] && (lb[j] > -DBL_MAX || ub[j] < DBL_MAX)) bounded = 1;
  return launch_sweep(m, kn, run, st, out, kf, bounded, stream);
}

#define HCHK(x)                                                                    \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      set_err("%s failed: %s", #x, hipGetErrorString(e_));                         \
      rc = FMCMC_ERR_DEVICE;                                                       \
      goto done;                                                                   \
    }                                                                              \
  } while (0)

int fmcmc_mcmc_run_host(const fmcmc_model* m, const fmcmc_kernel* kn, const fmcmc_run* run,
                        fmcmc_state* st, fmcmc_out* out, int device) {
  if (!m || !kn || !run || !st || !out) { set_err("null argument"); return FMCMC_ERR_ARG; }
  int rc = fmcmc_validate(m, kn, run);
  if (rc != FMCMC_OK) return rc;
  if (fmcmc_device_count() < 1) { set_err("no HIP device: the engine has no CPU fallback"); return FMCMC_ERR_DEVICE; }
  const int k = kn->k;
  const int kf = count_free(kn, kn->fixed);
  const int64_t C = run->nchains, S = fmcmc_kept_rows(run->nsteps, run->burnin, run->thin);
  const int64_t nwords = (run->nsteps + 31) / 32;
  const bool adaptive = (kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM);
  const bool mirror_h = (kn->kind == FMCMC_KERNEL_NMIRROR || kn->kind == FMCMC_KERNEL_UMIRROR);
  std::vector<void*> allocs;
  auto dalloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (bytes == 0) bytes = 8;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    allocs.push_back(p);
    return p;
  };
  fmcmc_model dm = *m;
  fmcmc_kernel dk = *kn;
  fmcmc_run dr = *run;
  fmcmc_state ds = *st;
  fmcmc_out dout = *out;
  hipStream_t stream = nullptr;
  int bounded = 0;
  for (int j = 0; j < k; j++)
    if (!kn->fixed[j] && (kn->lb[j] > -DBL_MAX || kn->ub[j] < DBL_MAX)) bounded = 1;

  HCHK(hipSetDevice(device));
  HCHK(hipStreamCreate(&stream));
#define UP(dst, src, bytes)                                                                  \
  do {                                                                                       \
    void* p_ = dalloc(bytes);                                                                \
    if (!p_) { set_err("hipMalloc(%zu) failed", (size_t)(bytes)); rc = FMCMC_ERR_DEVICE; goto done; } \
    if ((src) != nullptr) HCHK(hipMemcpyAsync(p_, (src), (bytes), hipMemcpyHostToDevice, stream)); \
    dst = (decltype(dst))p_;                                                                 \
  } while (0)
  if (m->p > 0) UP(dm.X, m->X, sizeof(double) * (size_t)m->p * m->n);
  UP(dm.y, m->y, sizeof(double) * (size_t)m->n);
  UP(dk.mu, kn->mu, sizeof(double) * k);
  UP(dk.scale, kn->scale, sizeof(double) * k);
  UP(dk.lb, kn->lb, sizeof(double) * k);
  UP(dk.ub, kn->ub, sizeof(double) * k);
  UP(dk.fixed, kn->fixed, (size_t)k);
  if (kn->scheme_seq && kn->scheme_len > 0) UP(dk.scheme_seq, kn->scheme_seq, sizeof(int32_t) * (size_t)kn->scheme_len);
  if (kn->constr && kn->kind == FMCMC_KERNEL_RAM) UP(dk.constr, kn->constr, sizeof(double) * (size_t)kf * kf);
  if (st->scheme_cols) UP(ds.scheme_cols, st->scheme_cols, sizeof(int32_t) * (size_t)C * run->nsteps);
  if (run->rng_mode == FMCMC_RNG_FED) {
    const int kz = variates_per_step(kn, kf);
    UP(dr.fed_logu, run->fed_logu, sizeof(double) * (size_t)C * run->nsteps);
    UP(dr.fed_z, run->fed_z, sizeof(double) * (size_t)C * run->nsteps * kz);
  }
  UP(ds.theta0, st->theta0, sizeof(double) * (size_t)C * k);
  UP(ds.f0, (double*)nullptr, sizeof(double) * (size_t)C);
  if (mirror_h) {
    if (!st->mirror_mu || !st->mirror_scale || !st->obs_arate || !st->abs_iter) {
      set_err("mirror kernels need state->mirror_mu, mirror_scale, obs_arate and abs_iter");
      rc = FMCMC_ERR_ARG;
      goto done;
    }
    UP(ds.abs_iter, st->fresh ? nullptr : st->abs_iter, sizeof(int64_t) * (size_t)C);
    UP(ds.mirror_mu, st->fresh ? nullptr : st->mirror_mu, sizeof(double) * (size_t)C * k);
    UP(ds.mirror_scale, st->fresh ? nullptr : st->mirror_scale, sizeof(double) * (size_t)C * k);
    UP(ds.obs_arate, st->fresh ? nullptr : st->obs_arate, sizeof(double) * (size_t)C);
  }
  if (adaptive) {
    UP(ds.abs_iter, st->fresh ? nullptr : st->abs_iter, sizeof(int64_t) * (size_t)C);
    UP(ds.Sigma, st->fresh ? nullptr : st->Sigma, sizeof(double) * (size_t)C * kf * kf);
    UP(ds.mean_prev, st->fresh ? nullptr : st->mean_prev, sizeof(double) * (size_t)C * kf);
    UP(ds.have_mean, st->fresh ? nullptr : st->have_mean, sizeof(int32_t) * (size_t)C);
    UP(ds.nerrors, (st->fresh || !st->nerrors) ? nullptr : st->nerrors, sizeof(int32_t) * (size_t)C);
    if (st->fresh || !st->nerrors) HCHK(hipMemsetAsync(ds.nerrors, 0, sizeof(int32_t) * (size_t)C, stream));
  }
  UP(dout.samples, (double*)nullptr, sizeof(double) * (size_t)C * k * S);
  HCHK(hipMemsetAsync(dout.samples, 0xff, sizeof(double) * (size_t)C * k * S, stream));  // NaN fill
  if (out->logpost) UP(dout.logpost, (double*)nullptr, sizeof(double) * (size_t)C * S);
  if (out->draws) UP(dout.draws, (double*)nullptr, sizeof(double) * (size_t)C * k * S);
  UP(dout.accept_count, (int64_t*)nullptr, sizeof(int64_t) * (size_t)C);
  if (out->accept_bits) UP(dout.accept_bits, (uint32_t*)nullptr, sizeof(uint32_t) * (size_t)C * nwords);
  UP(dout.status, (int32_t*)nullptr, sizeof(int32_t) * (size_t)C);
  UP(dout.status_step, (int64_t*)nullptr, sizeof(int64_t) * (size_t)C);
  UP(dout.status_theta, (double*)nullptr, sizeof(double) * (size_t)C * k);
  HCHK(hipMemsetAsync(dout.status_theta, 0, sizeof(double) * (size_t)C * k, stream));
#undef UP
  rc = launch_sweep(&dm, &dk, &dr, &ds, &dout, kf, bounded, stream);
  if (rc != FMCMC_OK) goto done;
#define DOWN(dst, src, bytes) HCHK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, stream))
  DOWN(st->theta0, ds.theta0, sizeof(double) * (size_t)C * k);
  DOWN(st->f0, ds.f0, sizeof(double) * (size_t)C);
  if (mirror_h) {
    DOWN(st->abs_iter, ds.abs_iter, sizeof(int64_t) * (size_t)C);
    DOWN(st->mirror_mu, ds.mirror_mu, sizeof(double) * (size_t)C * k);
    DOWN(st->mirror_scale, ds.mirror_scale, sizeof(double) * (size_t)C * k);
    DOWN(st->obs_arate, ds.obs_arate, sizeof(double) * (size_t)C);
  }
  if (adaptive) {
    DOWN(st->abs_iter, ds.abs_iter, sizeof(int64_t) * (size_t)C);
    DOWN(st->Sigma, ds.Sigma, sizeof(double) * (size_t)C * kf * kf);
    DOWN(st->mean_prev, ds.mean_prev, sizeof(double) * (size_t)C * kf);
    DOWN(st->have_mean, ds.have_mean, sizeof(int32_t) * (size_t)C);
    if (st->nerrors) DOWN(st->nerrors, ds.nerrors, sizeof(int32_t) * (size_t)C);
  }
  if (st->scheme_cols && run->rng_mode != FMCMC_RNG_FED && is_simple_kind(kn->kind) && kn->scheme == FMCMC_SCHEME_RANDOM)
    DOWN(st->scheme_cols, ds.scheme_cols, sizeof(int32_t) * (size_t)C * run->nsteps);
  DOWN(out->samples, dout.samples, sizeof(double) * (size_t)C * k * S);
  if (out->logpost) DOWN(out->logpost, dout.logpost, sizeof(double) * (size_t)C * S);
  if (out->draws) DOWN(out->draws, dout.draws, sizeof(double) * (size_t)C * k * S);
  DOWN(out->accept_count, dout.accept_count, sizeof(int64_t) * (size_t)C);
  if (out->accept_bits) DOWN(out->accept_bits, dout.accept_bits, sizeof(uint32_t) * (size_t)C * nwords);
  DOWN(out->status, dout.status, sizeof(int32_t) * (size_t)C);
  DOWN(out->status_step, dout.status_step, sizeof(int64_t) * (size_t)C);
  DOWN(out->status_theta, dout.status_theta, sizeof(double) * (size_t)C * k);
#undef DOWN
  HCHK(hipStreamSynchronize(stream));
  st->fresh = 0;
  for (int64_t c = 0; c < C; c++)
    if (out->status[c] != FMCMC_CHAIN_OK) {
      // message of R/mcmc.R:759-765
      set_err("fun(par) is undefined (chain %lld, status %d). Check either -fun- or the -lb- and -ub- "
              "parameters. This error ocurred during step i = %lld",
              (long long)(run->chain_base + c), out->status[c], (long long)out->status_step[c]);
      rc = FMCMC_ERR_CHAIN;
      break;
    }
done:
  for (void* p : allocs) hipFree(p);
  if (stream) hipStreamDestroy(stream);
  return rc;
}

}  // extern "C"
