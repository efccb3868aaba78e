// mh_common.hpp -- shared device code of the sweep kernels: launch arguments, LDS-only barriers, wave reductions on
// DPP / permlane swaps, reflection, per-chain LDS layout, streamed evaluation and the closed forms of the families.
// Included by mh_engine.hip only (one translation unit; everything lives in its anonymous namespace).
#pragma once
// Device code for gfx950 only: MFMA lane layouts, DPP / permlane forms, LDS sizes and -- for the wide kernels -- the
// behaviour of sc1 (write-through, L1-bypassing) accesses that the inter-workgroup hand-overs rest on are this target's.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fmcmc_amd device code is written for gfx950 (MI355X) only"
#endif


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
  const double* lg_hs;       // logistic: [intercept + p] data-only sums sum_i (y_i - 1/2) x_ij, then [p] largest |x| per column (logit_hs_kernel)
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
  double* obs_arate;         // [C][k] out (in when continuing)
  int bw;                    // adapt: window (0 = recursive)
  int hist_rows;             // adapt with bw > 0 / freq > 1: rows of the ring below (max(freq, bw - 1)), else 0
  double Sd;                 // adapt, bw > 0
  double* hist;              // [C][hist_rows][kf] ring of the last rows of ans[, which.] (row r in slot r % hist_rows)
  double until, eps, arate;
  double ram_df;             // ram: degrees of freedom of the Student-t variates (qfun), 0 = standard normal variates
  double ram_neg_exp;        // ram: eta(i, k) = min(1, k * exp(ram_neg_exp * log(i))), default -2/3
  const double* mu;
  const double* scale;
  const double* lb;
  const double* ub;
  const uint8_t* fixed;
  // run
  long long nchains, nsteps, burnin, thin, S, chain_base, step_base;
  long long ldS;             // row stride of the samples / draws columns and of logpost (>= S: fmcmc_out.ld_rows)
  unsigned long long seed;
  int rng_mode, fresh, ram_bounded, kz, tb, debug;
  // A long call of the stream-fed kernels (normal / uniform proposals: mh_sweep_mfma, mh_sweep_spec) runs as consecutive
  // STEP WINDOWS, each with a bounded materialised RNG stream (launch_sweep).  A continuation window is a launch whose
  // step 1 re-evaluates the state it starts from (same bits as the f0 it replaces) and whose steps 2.. are the call's steps
  // step_off + 2 ..: win_cont = 1 makes it take over the chain status and the partly filled word of the accept bitmap (the
  // accept COUNT is per launch: launch_sweep adds the windows up); thin_ctr0 is the thinning counter it starts with,
  // bits_stride the words per chain of the WHOLE call's bitmap, step_off what to add to a local step for status_step.
  int win_cont, thin_ctr0;
  long long bits_stride, step_off;
  long long nsteps_call;     // loop steps of the WHOLE call (row stride of scheme_cols, the plan of scheme = "random"; nsteps is a window's)
  double* win_sum;           // kernel_adapt in step windows: [C][kf] the running sum of this call's rows (its first running mean), carried from window to window; or NULL
  const double* fed_logu;
  const double* fed_z;
  const double* mf_stream;   // mh_sweep_mfma<.., EXT>: the observation slots beyond the operand registers, in operand order (mfma_build_stream)
  int mf_next;               //                          their number
  int spec_opt;              // mh_sweep_spec: observation slots per lane of this launch, rounded up to even (<= the instantiation's OPTMAX)
  int spec_cw;               // mh_sweep_spec: chains per workgroup of this launch: 4, or 2 / 1 (the latency form for few chains per GPU)
  // state
  double* theta0;
  double* f0;
  long long* abs_iter;
  double* Sigma;
  double* mean_prev;
  int* have_mean;
  int* nerrors;
  // observation-sharded evaluation (wide linear models, cooperative launch; 0 = off)
  int sh_ngrp;               // dataflow form (mh_sweep_wide2): chain groups, 2 or 4
  int sh_tiles;              // dataflow form: 1 = the uneven N-tile shares of its evaluator waves (default), 0 = the even split
  int shard;                 // canonical lanes per workgroup (512 / number of workgroups: 2 or 4)
  int sh_nslots;             // observations per canonical lane, ceil(n / 512); shard * sh_nslots <= SH_MAXO
  const double* sh_xs;       // [G][p][SH_MAXO] the workgroup's observations, column by column, slot-major (0 beyond n)
  const double* sh_ys;       // [G][SH_MAXO]
  double* sh_th;             // [k][nchains] proposals of all chains
  double* sh_part;           // [512][nchains] lane partials of all chains
  unsigned* sh_bar;          // barrier words, zeroed per launch (shard_barrier)
  const double* sh_mfma;     // [G][sh_mblk] the slices once more, in fp64-MFMA operand layout (shard_columns_mfma), or NULL
  int sh_mblk, sh_nmt;       // doubles per workgroup block; M-tiles of 16 observations per slice (1..3)
  int sh_t10;                // 1: the third M-tile holds 8 rows only and is laid out for two 4x4x4 MFMAs (shard_columns_mfma, T10)
  int sh_long;               // 1: the LONG-DATA form (shard_long): few chains, slices of thousands of observations; sh_xs = [G][p + 1][2 nslots]
  int sh_lcg;                //    chains whose residuals fit the workgroup's LDS block (sh_mblk doubles) at a time
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
__device__ __forceinline__ double readlane_d(double v, int src) {   // lane `src` (wave-uniform index) of v, as a scalar
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double uniform_d(double v);
__device__ __forceinline__ double sgpr_d(double v) {  // pin a wave-uniform double into an SGPR pair
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

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
// Lane N of the caller's 16-lane row in every lane of that row: ONE v_mov_b64_dpp row_newbcast.  (v_readlane_b32 twice gives
// a scalar pair whose fp64 consumers issue at ~6 cycles instead of ~4.5, and wait states between the two.)  N is an immediate:
// loops over rows go through static_for.
template <int N>
__device__ __forceinline__ double row_bcast(double v) {
  long long u = __double_as_longlong(v);
  u = __builtin_amdgcn_update_dpp(u, u, 0x150 + N, 0xf, 0xf, true);
  return __longlong_as_double(u);
}
// acc + bcast_N(bsrc) * other (NEG: acc - ...) as ONE instruction, v_fmac_f64 with the row broadcast as its DPP control -- a fused
// multiply-add like fmh_fma, the same bits.  (s_nop 1: a VGPR written by the VALU instruction in front needs two wait states before
// a DPP read, and the compiler's hazard recogniser does not look into inline assembly.)
template <int N, bool NEG = false>
__device__ __forceinline__ double fmac_row_bcast(double acc, double bsrc, double other) {
  if constexpr (NEG)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(other), "n"(N));
  else
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(other), "n"(N));
  return acc;
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in this order
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>());
  }
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

// v of the EVEN 16-lane row of each row pair, in both rows of the pair (v_permlane16_swap: no LDS round trip as __shfl_xor(v, 16)
// would make) / v of the upper 32 lanes in both halves (v_permlane32_swap)
__device__ __forceinline__ double even_row_d(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
}
__device__ __forceinline__ double upper_half_d(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
}

__device__ __forceinline__ double uniform_d(double v) { return sgpr_d(v); }

// Diagnostic build only (-DFMCMC_STAMP, tools/stamp_wide.py): s_memtime shares of the phases of a step, wave 0 of every
// workgroup, written over status_theta at the end of the sweep.  Compiled out of the product library.
struct Stamps { unsigned long long acc[16]; unsigned long long prev; };
#ifdef FMCMC_STAMP
__device__ __forceinline__ unsigned long long stamp_clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#ifndef FMCMC_STAMP_WAVE
#define FMCMC_STAMP_WAVE 0     /* -1: every wave stamps (tools/stamp_c5.py reads all eight) */
#endif
#define FMH_STAMP(S, i) do { if ((S) && (FMCMC_STAMP_WAVE < 0 || (int)(threadIdx.x >> 6) == FMCMC_STAMP_WAVE)) { const unsigned long long t_ = stamp_clk(); (S)->acc[i] += t_ - (S)->prev; (S)->prev = t_; } } while (0)
#else
#define FMH_STAMP(S, i) do { } while (0)
#endif

// ---- kernel_ram: update of the lower factor in product form (twin of the oracle's scan_sq_canon / ram_factor_update_canon;
// R/kernel_ram.R:136-146).  S (I + cp z z') S' = (S T)(S T)' with T = chol(I + sg p p') known in closed form (Gill, Golub,
// Murray & Saunders 1974): beta_0 = sg, beta_{j+1} = beta_j + p_j^2, T_jj = d_j = sqrt(beta_{j+1} / beta_j), T_ij = p_i p_j /
// (beta_j d_j).  So S'_ij = S_ij d_j + G_ij kappa_j, G_ij = sum_{m = j+1..i} S_im z_m, kappa_j = |cp| z_j / (beta_j d_j): the
// square root and the two divisions of a column are independent of every other column (lane = column: one of each per
// update instead of one dependent sqrt + two divisions PER COLUMN), and a row is two fma per element with G as its only
// carried value.  The sequential rank-1 update this replaces was 17.7 us of the 62 us C4 step (k = 50).
// Inclusive Hillis-Steele scan of q over the lanes, offsets 1, 2, 4, ...; lanes without a partner add +0 (q >= 0).
__device__ __forceinline__ double lane_scan_row16(double q) {   // k <= 16: inside one DPP row, row_shr with zero fill
  q = q + dpp_d<0x111>(q);
  q = q + dpp_d<0x112>(q);
  q = q + dpp_d<0x114>(q);
  q = q + dpp_d<0x118>(q);
  return q;
}
__device__ __forceinline__ double lane_scan_wave(double q) {    // k <= 64
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const double t = __shfl_up(q, s, 64);
    q = q + ((lane >= s) ? t : 0.0);
  }
  return q;
}
// lane j: d_j and kappa_j from cp, Pj = sum_{b<j} z_b^2, Pj1 = Pj + z_j^2; false when the column is unusable
__device__ __forceinline__ bool ram_coef(double cp, double Pj, double Pj1, double zj, double& d, double& kap) {
  const double acp = fmh_abs(cp), sg = (cp > 0.0) ? 1.0 : -1.0;
  const double b0 = fmh_fma(acp, Pj, sg), b1 = fmh_fma(acp, Pj1, sg);
  const double rho = b1 / b0;
  d = fmh_sqrt(rho);
  kap = (acp * zj) / (b0 * d);
  return (rho > 0.0) && fmh_isfinite(rho) && fmh_isfinite(kap);
}

// Split fp64 division.  The compiler expands a / b into v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal,
// q0 = a r, e = fma(-b, q0, a), v_div_fmas (= fma(e, r, q0)), v_div_fixup.  With both operands positive, normal and
// inside 2^-300 .. 2^300 the scale / fix-up instructions are identities, so r = div_recip(b) -- which depends on the
// denominator only and can be computed early -- followed by div_finish(a, b, r) is bit for bit a / b: three dependent
// instructions instead of twelve where the numerator arrives late.
__device__ __forceinline__ bool mfr_div_safe(double x) {
  return ((unsigned)(fmh_d2u(x) >> 32) - (723u << 20)) < (601u << 20);
}
__device__ __forceinline__ double div_recip(double b) {
  const double r0 = __builtin_amdgcn_rcp(b);
  const double r1 = fmh_fma(r0, fmh_fma(-b, r0, 1.0), r0);
  return fmh_fma(r1, fmh_fma(-b, r1, 1.0), r1);
}
__device__ __forceinline__ double div_finish(double a, double b, double r) {
  const double q0 = a * r;
  return fmh_fma(fmh_fma(-b, q0, a), r, q0);
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
  double* SigB;  // [kf*LD] adapt: Cholesky factor; ram: G, the partial sums of the proposal's S U chains
  double* sc;    // scalars: 0 f0, 1 f1
  double* mmu;   // [k] mirror kernels: adapted mean
  double* msc;   // [k] mirror kernels: adapted scale
};

__host__ __device__ inline int chain_lds_doubles(int k, int kf, int kind) {
  int LD = kf | 1;
  int mats = (kind == FMCMC_KERNEL_ADAPT || kind == FMCMC_KERNEL_RAM) ? 2 * kf * LD : 0;   // (ram: S, updated in place, and G)
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

// ---- logistic family: g(|eta|) = log(2 cosh(eta / 2)) off the row-polynomial table (include/fmh_detmath.h, fmh_logit_g):
// chain-vectorised twin of fmh_logit_g_scaled, the same operations in the same order for every element, so the same bits.
// The table (2400 rows x 6 coefficients, 115 KB) is staged in LDS by the logistic-only instantiations row by row (48 bytes:
// c0 c1 | c2 c3 | c4 c5) and a row is read with three ds_read_b128 from ONE address, 48 j, + immediate offsets 0 / 16 / 32
// (as three arrays of pairs the third array's offset, 76832, no longer fits the 16-bit offset field: one more vector add per
// lookup): 16-byte
// reads run at the full LDS rate (256 B/clk) with four waves per CU, 8-byte reads need four waves per SIMD for theirs
// (MI355X_MICROARCH.md, LDS), and tools/probe_logit_grid.hip measured the 8-byte form a third slower.  What a lookup costs
// beyond that is bank conflicts between the rows the lanes of one read hit: lanes = observations (the chain-sharded loops)
// scatter over the table, ~2.9-way; lanes = chains of one observation (the observation-sharded loop, logit_shard) hit equal or
// neighbouring rows, which neighbouring addresses serve without conflict.
typedef double lg_v2d __attribute__((ext_vector_type(2)));
constexpr int LG_LDS_DOUBLES = FMH_LG_ROWS * 6;       // rows of 48 bytes, c0 .. c5, as in include/fmh_logit_tab.h
constexpr int LG_LDS_TAIL = 2 + 6;                     // doubles allocated behind them: alignment slack + 12 control words (logit_shard's issue-priority turns)
__device__ __forceinline__ double* logit_table_align(double* p) {   // 16-byte aligned start inside [p, p + 2)
  return (double*)(((unsigned long long)p + 15ull) & ~15ull);
}
__device__ __forceinline__ void logit_stage_table(double* s_tab) {
  const double* t = fmh_lg_tab_();
  for (int i = threadIdx.x; i < FMH_LG_ROWS * 6; i += blockDim.x) s_tab[i] = t[i];
}
__device__ __forceinline__ unsigned cycle_stamp32() {   // low word of the shader clock
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return (unsigned)t;
}
// the 12 control words behind the table (LG_LDS_TAIL), for the launches that call logit_shard: "not set yet"
__device__ __forceinline__ void logit_reset_turns(double* s_tab) {
  if (threadIdx.x < 12) reinterpret_cast<unsigned*>(s_tab + LG_LDS_DOUBLES)[threadIdx.x] = 0xFFFFFFFFu;
}

// LDSTAB is a COMPILE-TIME property (the logistic-only instantiations always stage the table; the all-family kernels read
// the rows from global memory: scalar base + 32-bit lane offset).  us[] = 64 |eta| (>= 0 or NaN).
template <int CW, bool LDSTAB>
__device__ __forceinline__ void logit_g_vec(const double (&us)[CW], double (&out)[CW], const double* s_tab) {
  typedef const lg_v2d __attribute__((address_space(1))) * gptr2_t;
  typedef const lg_v2d __attribute__((address_space(3))) * lptr2_t;
  const gptr2_t gtab = (gptr2_t)(unsigned long long)fmh_lg_tab_();
  const lptr2_t ltab = (lptr2_t)s_tab;
  double s[CW];
  lg_v2d p0[CW], p1[CW], p2[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    s[c] = __builtin_amdgcn_fract(us[c]);
    unsigned j = (unsigned)us[c];                      // (saturating; NaN -> 0)
    j = (j < (unsigned)FMH_LG_ROWS) ? j : (unsigned)(FMH_LG_ROWS - 1);   // an element beyond the table is replaced below
    if constexpr (LDSTAB) { p0[c] = ltab[3u * j]; p1[c] = ltab[3u * j + 1u]; p2[c] = ltab[3u * j + 2u]; }
    else { p0[c] = gtab[3u * j]; p1[c] = gtab[3u * j + 1u]; p2[c] = gtab[3u * j + 2u]; }
  }
#pragma unroll
  for (int c = 0; c < CW; c++) {
    double q = fmh_fma(s[c], p2[c].y, p2[c].x);
    q = fmh_fma(s[c], q, p1[c].y);
    q = fmh_fma(s[c], q, p1[c].x);
    q = fmh_fma(s[c], q, p0[c].y);
    q = fmh_fma(s[c], q, p0[c].x);
    out[c] = (us[c] < (double)FMH_LG_ROWS) ? q : us[c] * FMH_LG_HALF_INV_SCALE;   // |eta| >= 37.5: |eta| / 2; NaN stays NaN
  }
}

// ---- observation-sharded evaluation of wide linear models (config C4) ---------------------------------------------
// With chains sharded over the CUs every CU streams all of X from L2 once per step, at the rate one CU gets out of its
// XCD's L2 (65-67 GB/s: 59 us per step at k = 50 -- the roofline of that design).  Here the OBSERVATIONS are sharded
// instead: workgroup b of G owns the canonical lanes b LPW .. b LPW + LPW - 1 (LPW = 512 / G) for ALL chains, i.e. a
// constant slice of LPW x nslots observations that it reads as scalar operands from a compact copy (15 KB at k = 50,
// stays in the scalar cache), and per step only the coefficients of all chains (k x nchains doubles) cross the chip.
// A step's evaluation is: every workgroup publishes the proposals of its own chains -> grid barrier -> thread = chain:
// the canonical fma chain over the columns for each observation of the slice, r^2 accumulated per lane in slot order ->
// lane partials of all chains to memory -> grid barrier -> thread = canonical lane again: it picks up its partial of
// the workgroup's own chains and the usual tree follows.  Same arithmetic, same order, same bits.
constexpr int SH_PAD = 32;             // row padding (doubles) of the exchange tables: a 4 KB row stride put every row
                                       // of a workgroup's strided accesses on the same L2 channel
constexpr int SH_MAXO = 40;            // observations of a slice held in registers by a thread; the slice (p x 40 doubles)
                                       // must stay in the 16 KB scalar cache: with 48 x 48 doubles (18 KB) every pass missed
// Grid barrier: 8 arrival counters + a top counter + 8 release words on separate cache lines, one lane per workgroup,
// relaxed agent-scope polling (5.3 us at 256 workgroups, tools/grid_barrier_custom.hip; cooperative_groups' grid.sync()
// takes 27 us).  NO agent-scope fence: its acquire half (buffer_inv sc1) drops the slice from the caches and the scalar
// loop then runs 3x slower (tools/scalar_slice_probe.hip: 11.8 -> 32.6 us per step), its release half writes back the
// whole L2.  Instead everything that crosses workgroups (coefficients, lane partials) is moved with agent-scope atomic
// loads and stores (sc1: write-through / L2-coherent per location), every thread drains its own stores (s_waitcnt
// vmcnt(0)) before the workgroup arrives, and workgroup-scope fences keep the compiler from moving accesses across.
// Cooperative launch guarantees co-residency; the spin is bounded anyway.
// (global address space on purpose: through a generic pointer these are FLAT instructions -- 64-bit VGPR addresses for
// every load in flight, and lgkmcnt shared with the scalar loads of the slice)
typedef double __attribute__((address_space(1))) * sh_gptr_t;
__device__ __forceinline__ void sh_store(double* p, double v) {
  __hip_atomic_store((sh_gptr_t)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two adjacent doubles (16-byte aligned) as ONE write-through store: an 8-byte sc1 store leaves L2 as a fabric write of
// its own, so the pair costs half the writes (MI355X_MICROARCH.md, stores of each flavour).  Not tracked by the compiler's
// wait-count insertion: every caller drains with an explicit s_waitcnt vmcnt(0) before it signals.
__device__ __forceinline__ void sh_store2(double* p, double v0, double v1) {
  typedef double d2v_t __attribute__((ext_vector_type(2)));
  const d2v_t v = {v0, v1};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double sh_load(const double* p) {
  return __hip_atomic_load((sh_gptr_t)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// fault: FMCMC_AMD_DEBUG=mode=512 (tests): workgroup 1 never arrives for epoch 3 and every spin gives up after ~25 ms instead
// of ~1 s -- the way a lost hand-over is EXERCISED rather than only argued (status 5, "results invalid", no hang)
__device__ __forceinline__ bool shard_barrier(unsigned* bar, unsigned epoch, bool fault = false) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);           // this thread's sc1 stores have been acknowledged
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
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
    unsigned spins = 0;
    const unsigned limit = fault ? 400000u : 20000000u;
    while (__hip_atomic_load(&rel[g * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > limit) { ok = false; break; }
    }
  }
  ok = !__syncthreads_or(ok ? 0 : 1);      // the verdict of thread 0, for every thread of the workgroup
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok;
}

// Step 2 of the sharded evaluation -- thread = chain: the slice's observations for that chain -- is a REAL function, not
// inlined into the sweep kernel: inlined, its 40 accumulators + the coefficient ring competed with everything the sweep
// keeps live across an evaluation, the instantiation sat at 256 VGPRs with 18-66 of them spilled, and the column loop
// ran at half the speed of the same loop alone (tools/scalar_slice_probe.hip).  Arguments arrive in VGPRs (the calling
// convention knows no uniform arguments), so everything uniform goes through v_readfirstlane first: without that the
// slice pointer is "divergent" and x comes through 40 vector loads per column instead of five scalar loads.
struct ShardCols {
  const double* xs;      // this workgroup's slice [p][SH_MAXO]
  const double* ys;      // [SH_MAXO]
  const double* th;      // [k][NC + SH_PAD] proposals of all chains
  double* part;          // [NC][NT + SH_PAD] lane partials
  long long n;
  int NC, p, ic, nslots, lane0, debug;
};
// 16 dwords, and it has to stay there: one more field -- even an unused one -- and the C4 step went from 33 to 41 us
// (the argument no longer travels in registers)
static_assert(sizeof(ShardCols) == 64, "ShardCols must stay at 16 dwords");
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long rfl_u64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// MAXO: observations of the slice that are actually walked (the first MAXO of its SH_MAXO slots; 20 when n <= 512 x 20 / LPW)
template <int LPW, int MAXO>
__device__ __attribute__((noinline)) void shard_columns(ShardCols c) {
  const int tid = threadIdx.x;
  const int NC = rfl_i(c.NC), NCP = NC + SH_PAD, p = rfl_i(c.p), ic = rfl_i(c.ic), nslots = rfl_i(c.nslots);
  const int lane0 = rfl_i(c.lane0), debug = rfl_i(c.debug);
  const long long n = (long long)rfl_u64((unsigned long long)c.n);
  // constant address space + uniform address = SCALAR loads (s_load_dwordx16: 8 observations per instruction); through
  // the generic pointer they were 48 broadcast vector loads per column and the evaluation took 96 us instead of ~15
  typedef const double __attribute__((address_space(4))) * cptr_t;
  const cptr_t xs = (cptr_t)rfl_u64((unsigned long long)c.xs);
  const cptr_t ys = (cptr_t)rfl_u64((unsigned long long)c.ys);
  const double* thg = (const double*)rfl_u64((unsigned long long)c.th);
  double* part = (double*)rfl_u64((unsigned long long)c.part);
  for (int cb = 0; cb < NC; cb += NT) {
    const int chain = cb + tid;
    const unsigned int chc = (unsigned int)(chain < NC ? chain : 0);
    double mu[MAXO];
    const double b0 = ic ? sh_load(thg + chc) : 0.0;
#pragma unroll
    for (int o = 0; o < MAXO; o++) mu[o] = b0;
    // coefficients in blocks of 8, THREE blocks in flight: they were written by other XCDs a barrier ago, every load is
    // an L2 miss of 1-3 us and a block's 320 FMAs cover ~1 us; one block ahead stalled on every block (36 us for the
    // columns instead of 12), all of them at once (6 blocks + tail) spilled 300 registers
    constexpr int JB8 = 8, RING = 3;
    const int pe = (debug & 64) ? 1 : p;
    double tb[RING][JB8];
    // (column index clamped instead of a guard per load: 32 guarded loads compiled to 32 branches)
#define SH_LOAD_BLOCK(q, j0)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < JB8; u++) {                                                          \
      const int jj = ((j0) + u < pe) ? (j0) + u : pe - 1;                                                      \
      tb[q][u] = sh_load(thg + ((unsigned int)((ic + jj) * NCP) + chc));                                       \
    }
#pragma unroll
    for (int q = 0; q < RING; q++) { SH_LOAD_BLOCK(q, q * JB8) }
    for (int jb = 0; jb < pe; jb += RING * JB8) {
#pragma unroll
      for (int q = 0; q < RING; q++) {
        const int j0 = jb + q * JB8;
        if (j0 < pe) {
#pragma unroll
          for (int u = 0; u < JB8; u++) {
            if (j0 + u < pe) {               // uniform (and on purpose: without a guard between the columns the
              const cptr_t xc = xs + (j0 + u) * SH_MAXO;   // scheduler hoists scalar loads until 700 B per lane spill)
#pragma unroll
              for (int o = 0; o < MAXO; o++) mu[o] = fmh_fma(xc[o], tb[q][u], mu[o]);
            }
          }
          if (j0 + RING * JB8 < pe) { SH_LOAD_BLOCK(q, j0 + RING * JB8) }
        }
      }
    }
#undef SH_LOAD_BLOCK
    double al[LPW];
#pragma unroll
    for (int q = 0; q < LPW; q++) al[q] = 0.0;
#pragma unroll
    for (int o = 0; o < MAXO; o++) {             // o = slot * LPW + lane-in-slice: slot order per lane
      const int sl = o / LPW, q = o % LPW;
      const bool valid = sl < nslots && ((long long)(lane0 + q) + (long long)NT * sl) < n;   // uniform
      const double r = valid ? ys[o] - mu[o] : 0.0;                      // fma(0, 0, acc) == acc exactly
      al[q] = fmh_fma(r, r, al[q]);
    }
    if (chain < NC) {
#pragma unroll
      for (int q = 0; q < LPW; q++) sh_store(&part[(long long)chain * (NT + SH_PAD) + lane0 + q], al[q]);
    }
  }
}

// ---- step 2 on the matrix cores: mu[observation][chain] = b0 + X_slice . B as v_mfma_f64_16x16x4 tiles -------------------
// One tile = 16 observations (M) x 16 chains (N) x 4 columns (K).  The instruction is BITWISE
// fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))) per output (tools/mfma64_16x16_exact.hip), so K-blocks chained
// through C in ascending column order ARE the canonical fma chain of an observation: same bits as shard_columns, the
// streamed kernels and the oracle.  Operand layout (lane l): A[i = l % 16][kk = l / 16], B[kk = l / 16][j = l % 16],
// D register r = row 4 r + l / 16, column l % 16 (probed).  So a lane GROUP g = l / 16 ends up with 4 rows per M-tile for chain
// l % 16, and the slice is laid out (shard_build_mfma, mh_engine.hip) so that those rows are CONSECUTIVE SLOTS of one
// canonical lane: group g works for canonical lane q = g / H of the slice (H = 4 / LPW groups per lane), its value
// t = 4 mt + r (M-tile mt, register r) is slot spg h + t, h = g % H, spg = ceil(nslots / H) <= 10.  The r^2 chain of a
// canonical lane therefore runs inside a lane in slot order; with H = 2 the second half continues from the partner
// group's value (lane ^ 16).  Rows / columns beyond the data are 0 in A and masked out of the chain.
// LDS block per workgroup (doubles), staged once per launch: [0, 32) validity bits (u32 per lane, bit t), [32, 32 + 12 x 64)
// y in D layout [t][lane], then the A tiles [mt][kb][lane].
// T10 (three M-tiles of which the third carries values t = 8, 9 only, i.e. 8 of its 16 rows -- config C4: 20 slots per lane,
// 10 per lane group): a 16x16x4 tile would spend 64 matrix-core cycles per K-block on 8 rows of zeros.  Its two live D
// registers are computed by two v_mfma_f64_4x4x4_4b instead (16.7 cycles each): that instruction's four blocks take the SAME
// B register (B at lane 16 kk + 4 blk + j = chain 4 blk + j of the N-tile, which is where the 16x16x4 form keeps chain
// l % 16), deliver D at lane 16 i + 4 blk + j = row i of chain 4 blk + j, which is the 16x16x4 form's D register r for rows
// 4 r + l / 16, and it is the same fma chain bit for bit (mh_mfma.hpp, tools/mfma64_exact.hip).  Only A differs: lane
// 16 kk + 4 blk + i wants row 4 r + i of the tile whatever blk is, so the third tile's region holds, per K-block, the 32
// doubles [kk][i][r] (one ds_read_b128 per lane and K-block) instead of 64.  Matrix-core time per tile: 12 x (2 x 64 + 2 x 16.7)
// = 1936 cycles against 2304.
constexpr int SHM_T = 24;                       // D values per lane and N-tile at most (6 M-tiles x 4 registers; 3 until round 4: n <= 10,240)
// doubles in front of the A tiles of a block with nmt M-tiles: 32 of validity bits, then y in D layout for its 4 nmt values per lane
// (never less than the three-tile header of rounds 2-3, so that the tuned forms keep their addresses)
__host__ __device__ constexpr int shm_hdr(int nmt) { return 32 + (nmt < 3 ? 12 : 4 * nmt) * 64; }
constexpr int SHM_KBMAX = 16;                   // K-blocks of 4 columns: p <= 64
constexpr int SHM_T10_FULL = 7;                 // form T10: values 0..6 of every lane group are observations (checked by the host)
typedef double d4_t __attribute__((ext_vector_type(4)));
struct ShardMfma {
  const double* th;      // [k][ncp] proposals of all chains
  double* part;          // [NC][NT + SH_PAD] lane partials
  unsigned lds;          // LDS address of the block
  int NC, p, ic, lane0, tcount;  // NC: chains of the set this call evaluates; tcount: at most this many tiles (0: no limit)
  int ncp;               // row stride of th (all chains of the launch + SH_PAD)
  int cstride, coff;     // chain of the set's member l: cstride * l + coff (1, 0: all chains; 2, g: chain group g of mh_sweep_wide2)
  int thoff;             // column of member 0 in a row of th; the set's members are CONTIGUOUS there (0: all chains; g NH: group g)
  int tfirst, tstep;     // N-tiles of the calling wave: tfirst, tfirst + tstep, ...
};
static_assert(sizeof(ShardMfma) <= 64, "ShardMfma must travel in registers (16 dwords)");
// KBC > 0: the number of K-blocks is the compile-time constant KBC (config C4: 12).  With a run-time count every K-block is
// a basic block of its own -- a branch, reloads of spilled scalars, and nothing of one block scheduled into the next: a
// lone wave ran a tile in 2.6 us (tools/exp_shard_mfma.hip), 1.9 us with the count known, 1.04 us being its matrix-core time.
template <int LPW, int NMT, int KBC = 0, bool T10 = false>
__device__ __attribute__((noinline)) void shard_columns_mfma(ShardMfma c) {
  static_assert(!T10 || (NMT == 3 && KBC > 0), "T10 is the form of three M-tiles with a compile-time K-block count");
  constexpr int NM16 = T10 ? 2 : NMT;           // M-tiles computed as 16x16x4
  constexpr int NTV = T10 ? 10 : 4 * NMT;       // D values per lane
  // (T10) values t < SHM_T10_FULL are observations in EVERY lane of every workgroup (the host takes the form only then:
  // slots spg h + t <= nslots - 2 are full), so their residuals need no mask: 14 vector instructions less per tile
  constexpr int NFULL = T10 ? SHM_T10_FULL : 0;
  const int lane = threadIdx.x & 63;
  const int NC = rfl_i(c.NC), NCP = rfl_i(c.ncp), p = rfl_i(c.p), ic = rfl_i(c.ic), lane0 = rfl_i(c.lane0);
  const int cstride = rfl_i(c.cstride), coff = rfl_i(c.coff), thoff = rfl_i(c.thoff), tfirst = rfl_i(c.tfirst), tstep = rfl_i(c.tstep);
  const int tcount = rfl_i(c.tcount);
  const int KB = KBC > 0 ? KBC : (p + 3) >> 2;
  const double* thg = (const double*)rfl_u64((unsigned long long)c.th);
  double* part = (double*)rfl_u64((unsigned long long)c.part);
  typedef __attribute__((address_space(3))) const double* ldsc_t;
  typedef __attribute__((address_space(3))) const unsigned* ldsu_t;
  const unsigned lbase = (unsigned)rfl_i((int)c.lds);
  const ldsc_t blk = (ldsc_t)(unsigned long long)lbase;
  const unsigned vmask = ((ldsu_t)(unsigned long long)lbase)[lane];
  double ya[SHM_T];
#pragma unroll
  for (int t = 0; t < NTV; t++) ya[t] = blk[32 + 64 * t + lane];
  constexpr int HDR = shm_hdr(NMT);
  const ldsc_t xa = blk + HDR + lane;           // tile (mt, kb) at xa[(mt KB + kb) 64]
  const int kk = lane >> 4, j = lane & 15;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const d2_t* ldsc2_t;
  // (T10) rows 4 r + i of the third tile, r = 0, 1, for column kk of K-block kb: xa4[kb * 32]
  const ldsc2_t xa4 = (ldsc2_t)(blk + HDR + 2 * KB * 64 + 2 * (4 * kk + (lane & 3)));
  const int ntiles_all = (NC + 15) >> 4;
  // (the wave's share: tfirst, tfirst + tstep, ..., at most tcount of them)
  const int ntiles = (tcount > 0 && tfirst + tcount * tstep < ntiles_all) ? tfirst + (tcount - 1) * tstep + 1 : ntiles_all;
  // B operand of N-tile T: coefficient column 4 kb + kk of chain 16 T + j; C operand: its intercept.  Written by other
  // XCDs a barrier ago: every load is a miss of 1-3 us, so the next tile's are in flight under this tile's MFMAs.
  double Bc[SHM_KBMAX], Bn[SHM_KBMAX], c0c = 0.0, c0n = 0.0;
  const char* thb = (const char*)thg;
  // (KBC) the rows' base addresses as opaque SCALAR values, so that a load is `global_load v, voffset, s[base]` with no vector
  // address arithmetic at all: the fp64 datapath a tile's ~100 vector instructions run on is the one its MFMAs run on
  unsigned long long rowb[KBC > 0 ? KBC : 1];
  if constexpr (KBC > 0) {
#pragma unroll
    for (int kb = 0; kb < KBC; kb++) {
      rowb[kb] = (unsigned long long)thb + (unsigned long long)(4 * kb) * (unsigned long long)NCP * 8ull;
      asm volatile("" : "+s"(rowb[kb]));
    }
  }
// (addresses as a wave-uniform row pointer + a 32-bit lane offset in bytes: one global_load with an SGPR base per operand;
//  as 64-bit lane arithmetic every load cost three more vector instructions, 0.17 us per tile for the 13 of them)
#define SHM_LOAD_B(T_, B_, c0_)                                                                          \
  {                                                                                                      \
    const int ch_ = 16 * (T_) + j;                                                                       \
    const unsigned int chc_ = (unsigned int)(thoff + (ch_ < NC ? ch_ : NC - 1));                         \
    const unsigned int cb_ = chc_ * 8u, rb_ = (unsigned int)((ic + kk) * NCP) * 8u + cb_;                \
    c0_ = ic ? sh_load((const double*)(thb + cb_)) : 0.0;                                                \
    _Pragma("unroll") for (int kb = 0; kb < (KBC > 0 ? KBC : SHM_KBMAX); kb++) {                         \
      if (kb < KB) {                                                                                     \
        const int col_ = 4 * kb + kk;                                                                    \
        if (KBC > 0 && kb < KBC - 1)                                                                     \
          B_[kb] = sh_load((const double*)((const char*)rowb[kb] + rb_));                                \
        else                                                                                             \
          B_[kb] = sh_load(thg + ((unsigned int)((ic + (col_ < p ? col_ : p - 1)) * NCP) + chc_));       \
      }                                                                                                  \
    }                                                                                                    \
  }
#pragma unroll
  for (int kb = 0; kb < SHM_KBMAX; kb++) { Bc[kb] = 0.0; Bn[kb] = 0.0; }
  // Where the wave WAITS for a tile's operands is pinned by hand (SHM_PIN: an empty asm that reads the registers, so the
  // compiler's s_waitcnt lands in front of it and the values are plain registers afterwards): the first tile's right behind
  // their loads, the next tile's at the end of the current one, BEFORE its store.  Left to the compiler's wait-count
  // bookkeeping, the loads of tile T + 1 (issued behind a branch) met "none issued" at the join, which put s_waitcnt vmcnt(0)
  // in front of tile T's fifth MFMA: every tile waited out the full latency of the next tile's loads and the prefetch hid
  // nothing (lone wave: 1.59 us per tile against 0.83 us of matrix-core time).  With the wait pinned at the end of the tile
  // -- in front of the store, not behind it: vmcnt counts the store too -- the loads may sit behind a branch again (the pin's
  // wait is then the conservative vmcnt(0), which at that place is what is wanted): the last tile of a visit loads nothing.
#define SHM_PIN(B_, c0_)                                                                                 \
  {                                                                                                      \
    asm volatile("" : "+v"(c0_));                                                                        \
    _Pragma("unroll") for (int kb = 0; kb < (KBC > 0 ? KBC : SHM_KBMAX); kb++) asm volatile("" : "+v"(B_[kb])); \
  }
  int T = tfirst;                                // N-tiles of this wave
#ifndef SHM_DEPTH
#define SHM_DEPTH 1
#endif
  // The register sets of the operands change roles from tile to tile (the loop below is unrolled) instead of being copied: a copy
  // was 13 v_mov_b64 per tile, and every vector instruction of a tile runs on the fp64 datapath its MFMAs need (a tile is 1920
  // matrix-core cycles + ~4.4 per vector instruction: 105 -> 66 of them took the harness from 6.1 to 5.5 us per visit of a pair).
  // SHM_DEPTH = 2 (three sets, the loads of tile T + 2 issued at the start of tile T; compile-time knob) was measured at C4 and
  // dropped: 17.4 us per step against 16.7 with loads issued only for tiles that exist, 19.1 with the unconditional form -- a
  // visit's loads all at once are a burst of 20 MB of sc1 reads from every workgroup at the same moment.
  double Bx[SHM_KBMAX], c0x = 0.0;
#pragma unroll
  for (int kb = 0; kb < SHM_KBMAX; kb++) Bx[kb] = 0.0;
  if (T >= ntiles) return;                       // (a wave without a tile)
  SHM_LOAD_B(T, Bc, c0c)
  if (SHM_DEPTH == 2 && T + tstep < ntiles) SHM_LOAD_B(T + tstep, Bn, c0n)
  SHM_PIN(Bc, c0c)
  // one tile: operands in (Bc_, c0c_); the next tile's are (being) loaded into (Bn_, c0n_) and waited for at the end of this one;
  // (Bl_, c0l_) is the set this tile issues loads into (depth 2: the tile after the next; depth 1: the next = Bn_)
  auto one_tile = [&](double (&Bc_)[SHM_KBMAX], double& c0c_, double (&Bn_)[SHM_KBMAX], double& c0n_,
                      double (&Bl_)[SHM_KBMAX], double& c0l_) __attribute__((always_inline)) {
#ifndef SHM_UNCOND_LOAD
    if (T + SHM_DEPTH * tstep < ntiles) SHM_LOAD_B(T + SHM_DEPTH * tstep, Bl_, c0l_)
#else
    SHM_LOAD_B((T + SHM_DEPTH * tstep < ntiles ? T + SHM_DEPTH * tstep : T), Bl_, c0l_)
#endif
    d4_t acc[NM16];
#pragma unroll
    for (int mt = 0; mt < NM16; mt++) acc[mt] = (d4_t){c0c_, c0c_, c0c_, c0c_};
    double acc4[2] = {c0c_, c0c_};                 // (T10) D registers 0, 1 of the third tile
    double a_cur[NM16], a_nxt[NM16];
    d2_t a4_cur = (d2_t){0.0, 0.0}, a4_nxt = (d2_t){0.0, 0.0};
#pragma unroll
    for (int mt = 0; mt < NM16; mt++) a_cur[mt] = xa[(mt * KB) * 64];
    if constexpr (T10) a4_cur = xa4[0];
#pragma unroll
    for (int kb = 0; kb < (KBC > 0 ? KBC : SHM_KBMAX); kb++) {
      if (kb < KB) {
        const int kn = (kb + 1 < KB) ? kb + 1 : kb;
#pragma unroll
        for (int mt = 0; mt < NM16; mt++) a_nxt[mt] = xa[(mt * KB + kn) * 64];
        if constexpr (T10) a4_nxt = xa4[kn * 32];
        // (a padded column must not turn an infinite coefficient into NaN; with KBC only the last block can hold one)
        const double b = (KBC > 0 && kb < KBC - 1) ? Bc_[kb] : ((4 * kb + kk < p) ? Bc_[kb] : 0.0);
#pragma unroll
        for (int mt = 0; mt < NM16; mt++) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[mt], b, acc[mt], 0, 0, 0);
        if constexpr (T10) {
          acc4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4_cur[0], b, acc4[0], 0, 0, 0);
          acc4[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4_cur[1], b, acc4[1], 0, 0, 0);
        }
        if constexpr (KBC > 0) {   // the block's other instructions BETWEEN its MFMAs: a wave issues in order, and an MFMA
#pragma unroll                     // holds the issue port until the matrix core takes it, 64 cycles after the previous one
          for (int mt = 0; mt < NM16; mt++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          if constexpr (T10) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < NM16; mt++) a_cur[mt] = a_nxt[mt];
        if constexpr (T10) a4_cur = a4_nxt;
      }
    }
    double rr[NTV];
#pragma unroll
    for (int t = 0; t < NTV; t++) {
      const double mu = (T10 && t >= 8) ? acc4[t & 1] : acc[(T10 && t >= 8) ? 0 : (t >> 2)][t & 3];
      rr[t] = (t < NFULL || ((vmask >> t) & 1u)) ? ya[t] - mu : 0.0;   // fma(0, 0, a) == a exactly
    }
    double a = 0.0;
#pragma unroll
    for (int t = 0; t < NTV; t++) a = fmh_fma(rr[t], rr[t], a);
    const int lch = 16 * T + j;                  // member of the set; its chain:
    const int chain = cstride * lch + coff;
    asm volatile("" : "+v"(a), "+v"(c0n_));       // (behind the tile's last MFMA: the pins below follow this one in order,
    __builtin_amdgcn_sched_barrier(0);           //  and the scheduler must not hoist the register copies they imply)
    SHM_PIN(Bn_, c0n_)
    if constexpr (LPW == 2) {
      double a2 = even_row_d(a);                  // groups 1 and 3 continue where groups 0 and 2 stopped (what 0 and 2 make of it is not used)
#pragma unroll
      for (int t = 0; t < NTV; t++) a2 = fmh_fma(rr[t], rr[t], a2);
      const double a_hi = upper_half_d(a2);       // canonical lane 1 of the slice (group 3) next to lane 0 (group 1)
      if (kk == 1 && lch < NC) {
        double* dst = &part[(long long)chain * (NT + SH_PAD) + lane0];
        sh_store2(dst, a2, a_hi);
      }
    } else {
      if (lch < NC) sh_store(&part[(long long)chain * (NT + SH_PAD) + lane0 + kk], a);
    }
  };
  if (SHM_DEPTH == 2) {
    while (T < ntiles) {
      one_tile(Bc, c0c, Bn, c0n, Bx, c0x); T += tstep;
      if (T >= ntiles) break;
      one_tile(Bn, c0n, Bx, c0x, Bc, c0c); T += tstep;
      if (T >= ntiles) break;
      one_tile(Bx, c0x, Bc, c0c, Bn, c0n); T += tstep;
    }
  } else {
    while (T < ntiles) {
      one_tile(Bc, c0c, Bn, c0n, Bn, c0n); T += tstep;
      if (T >= ntiles) break;
      one_tile(Bn, c0n, Bc, c0c, Bc, c0c); T += tstep;
    }
  }
}

#undef SHM_LOAD_B
#undef SHM_PIN

// The slice product with the K-block count a compile-time constant where one exists (two / three M-tiles of two canonical lanes per
// workgroup -- up to six beyond n = 10,240 --, 4 .. 16 K-blocks: p = 13 .. 64): config C4's width had it since round 2 (12 K-blocks: 25.4 -> 20.1 us per step); at the
// other widths a tile ran the run-time loop -- p = 30, n = 1e4, 512 chains: 21.0 -> 17.5 us per step with KBC = 8.
template <int LPW, int NMT>
__device__ __forceinline__ void shard_mfma_dispatch(const ShardMfma& sm, int KB, int t10) {
  if constexpr (LPW == 2 && NMT >= 2) {
    if (NMT == 3 && KB == 12 && t10) { shard_columns_mfma<2, 3, 12, true>(sm); return; }
    // (four to six M-tiles: up to 12 K-blocks -- beyond, the unrolled tile spills: p = 60, n = 2e4, 64 chains 16.4 -> 25.6 us per step)
    if (NMT <= 3 || KB <= 12)
    switch (KB) {
#define SHM_KB(K_) case K_: shard_columns_mfma<2, NMT, K_>(sm); return;
      SHM_KB(4) SHM_KB(5) SHM_KB(6) SHM_KB(7) SHM_KB(8) SHM_KB(9) SHM_KB(10) SHM_KB(11) SHM_KB(12) SHM_KB(13) SHM_KB(14) SHM_KB(15) SHM_KB(16)
#undef SHM_KB
      default: break;
    }
  } else if constexpr (NMT == 3) {
    if (KB == 12 && t10) { shard_columns_mfma<LPW, 3, 12, true>(sm); return; }
    if (KB == 12) { shard_columns_mfma<LPW, 3, 12>(sm); return; }
  }
  shard_columns_mfma<LPW, NMT>(sm);
}

// ---- LONG-DATA form of the sharded evaluation (round 4; tools/dispatch_audit.py: four chains are ONE workgroup on the chain-sharded
// kernels -- n = 1e5 observations took 34 us per step on one CU, 255 CUs idle).  Workgroup b owns the canonical lanes 2b, 2b + 1 as
// in the other sharded forms, but its slice is long (n / 256 observations) and the chains are few: (a) all 512 threads compute the
// residuals r = y - (b0 + x b) of the slice for a group of chains -- the fma chain over the columns in the canonical order -- into
// LDS; (b) ONE thread per (chain, canonical lane) walks its residuals in slot order, acc = fma(r, r, acc): the canonical lane sum is
// a sequential chain, ~8 cycles per slot whatever is done, and that walk IS the evaluation time (n = 1e6: 1953 slots, ~7 us).
struct ShardLong {
  const double* xs;      // this workgroup's slice [p + 1][nobs]: columns, then y; observation o = 2 slot + q (0 beyond n)
  const double* th;      // [k][ncp] proposals of all chains
  double* part;          // [NC][NT + SH_PAD] lane partials
  unsigned lds;          // LDS address of the term block [lcg][2][shard_long_row(nslots)] + [lcg][SHL_BS] coefficients
  unsigned tab;          // logistic: LDS address of the g table
  int n;                 // observations (< 2^31: the host checks)
  int NC, ncp, p, ic, nslots, lane0, lcg;
};
static_assert(sizeof(ShardLong) <= 64, "ShardLong must travel in registers (16 dwords)");
// slots of a (chain, canonical lane) row of the residual block: nslots rounded up to the walk's block of 16, plus one block the
// walk's prefetch may touch
__host__ __device__ constexpr int shard_long_row(int nslots) { return ((nslots + 15) & ~15) + 16; }
// FAM: Gaussian linreg -- terms r = y - (b0 + x b), lane sum acc = fma(r, r, acc) -- or logistic -- terms g(|64 eta|) off the table
// in LDS (the checked form of logit_shard's term: any |eta|), lane sum acc = acc + g; up to 16 covariates.
constexpr int SHL_BS = 64;     // doubles per chain of the coefficient copies in LDS (k <= 64)
template <int FAM, int PMAX /* columns held in registers per observation: 16, or 62 for wide linear models */>
__device__ __attribute__((noinline)) void shard_long(ShardLong c) {
  typedef __attribute__((address_space(3))) double* ldsd_t;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr bool LG = FAM == FMCMC_FAM_LOGISTIC;
  const int tid = threadIdx.x;
  const int NC = rfl_i(c.NC), NCP = rfl_i(c.ncp), p = rfl_i(c.p), ic = rfl_i(c.ic), nslots = rfl_i(c.nslots), lane0 = rfl_i(c.lane0), lcg = rfl_i(c.lcg);
  const long long n = (long long)rfl_i(c.n);
  const double* xs = (const double*)rfl_u64((unsigned long long)c.xs);
  const double* thg = (const double*)rfl_u64((unsigned long long)c.th);
  double* part = (double*)rfl_u64((unsigned long long)c.part);
  const double* s_tab = (const double*)(ldsd_t)(unsigned long long)(unsigned)rfl_i((int)c.tab);
  const int nobs = 2 * nslots, nb = ic + p, NSP = shard_long_row(nslots), nwalk = NSP - 16;
  double* s_r = (double*)(ldsd_t)(unsigned long long)(unsigned)rfl_i((int)c.lds);   // [lcg][2][NSP] residuals, a row per (chain, lane)
  double* s_b = s_r + (long long)lcg * 2 * NSP;                                       // [lcg][SHL_BS] coefficients of the group's chains (logistic: times 64, exact)
  for (int c0 = 0; c0 < NC; c0 += lcg) {
    const int ncg = (NC - c0 < lcg) ? NC - c0 : lcg;
    for (int idx = tid; idx < ncg * nb; idx += NT) {
      const int cc = idx / nb, j = idx - cc * nb;
      const double bv = sh_load(thg + ((unsigned int)(j * NCP) + (unsigned int)(c0 + cc)));
      s_b[cc * SHL_BS + j] = LG ? bv * FMH_LG_SCALE : bv;
    }
    lds_barrier();
    // (a) residuals of the slice, every thread a stride of its observations; slots beyond the data (and the padding of a row up to
    //     the walk's block) hold 0: fma(0, 0, acc) == acc exactly, so the walk below needs no tail
    for (int o = tid; o < 2 * nwalk; o += NT) {
      const long long i = (long long)NT * (o >> 1) + lane0 + (o & 1);
      const bool valid = o < nobs && i < n;
      const int oc = o < nobs ? o : nobs - 1;
      double x[PMAX];
#pragma unroll
      for (int j = 0; j < PMAX; j++) x[j] = (j < p) ? xs[(long long)j * nobs + oc] : 0.0;
      const double yv = LG ? 0.0 : xs[(long long)p * nobs + oc];
      double* dst = s_r + (long long)(o & 1) * NSP + (o >> 1);
      for (int cc = 0; cc < ncg; cc++) {
        const double* bj = s_b + cc * SHL_BS;
        double m = ic ? bj[0] : 0.0;
#pragma unroll
        for (int j = 0; j < PMAX; j++) if (j < p) m = fmh_fma(x[j], bj[ic + j], m);
        double term;
        if constexpr (LG) {
          const double us1[1] = {__builtin_fabs(m)};
          double g1[1];
          logit_g_vec<1, true>(us1, g1, s_tab);
          term = g1[0];
        } else {
          term = yv - m;
        }
        dst[(long long)cc * 2 * NSP] = valid ? term : 0.0;
      }
    }
    lds_barrier();
    // (b) the canonical lane sum: ONE thread per (chain, lane) walks its row in slot order -- blocks of 16 slots, the next block's
    //     eight ds_read_b128 in flight under this block's sixteen dependent FMAs
    if (tid < 2 * ncg) {
      const d2_t* rp = reinterpret_cast<const d2_t*>(s_r + (long long)tid * NSP);   // (row cc * 2 + q == tid)
      d2_t cur[8], nxt[8];
#pragma unroll
      for (int u = 0; u < 8; u++) cur[u] = rp[u];
      double acc = 0.0;
      for (int sl = 0; sl < nwalk; sl += 16) {
#pragma unroll
        for (int u = 0; u < 8; u++) nxt[u] = rp[(sl >> 1) + 8 + u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if constexpr (LG) { acc = acc + cur[u].x; acc = acc + cur[u].y; }      // (a slot without an observation adds +0: exact)
          else { acc = fmh_fma(cur[u].x, cur[u].x, acc); acc = fmh_fma(cur[u].y, cur[u].y, acc); }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) cur[u] = nxt[u];
      }
      sh_store(&part[(long long)(c0 + (tid >> 1)) * (NT + SH_PAD) + lane0 + (tid & 1)], acc);
    }
    lds_barrier();
  }
}

// lane partials acc[c] of canonical lane `tid` for the CW chains of this workgroup, via the sharded evaluation
// (the logistic family's step 2, logit_shard below, is reached through this forward-declared hook)
template <int LPW>
__device__ __forceinline__ void eval_sharded_logit_step(const SweepArgs& A, const double* s_tab);
template <int CW, int LPW, int FAM = FMCMC_FAM_GAUSSIAN_LINREG>
__device__ __forceinline__ bool eval_sharded(const SweepArgs& A, double* const* th, double (&acc)[CW], unsigned& epoch,
                                             const double* s_mblk /* LDS block of the MFMA form (logistic: the g table), or NULL */, Stamps* stp = nullptr) {
  const int tid = threadIdx.x;
  const int NC = (int)A.nchains, NCP = NC + SH_PAD, p = A.p, ic = A.intercept, nb = ic + p;
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  // 1. publish the coefficients of this workgroup's chains, [coefficient][chain]
  for (int idx = tid; idx < CW * nb; idx += NT) {
    const int c = idx / nb, j = idx - c * nb;
    if (c < ncw) sh_store(&A.sh_th[(long long)j * NCP + cg0 + c], th[c][j]);
  }
  // (debug bits 32 / 64 / 128: timing ablations.  A hand-over that timed out is not waited for again: bit 31 of the
  //  epoch marks the sweep as lost, its remaining steps run through without barriers and the host raises.)
  constexpr unsigned LOST = 0x80000000u;
  bool ok = true;
  FMH_STAMP(stp, 3);
  if (!(A.debug & 32) && !(epoch & LOST)) { ok = shard_barrier(A.sh_bar, ++epoch, (A.debug & 512) != 0); if (!ok) epoch |= LOST; }
  FMH_STAMP(stp, 4);
  // 2. thread = chain: the slice's observations for that chain
  if (A.sh_long) {   // few chains on long data (both families)
    typedef __attribute__((address_space(3))) const double* ldsc_t;
    constexpr bool LG = FAM == FMCMC_FAM_LOGISTIC;
    ShardLong sl;
    sl.xs = A.sh_xs + (long long)blockIdx.x * (p + 1) * 2 * A.sh_nslots;
    sl.th = A.sh_th; sl.part = A.sh_part; sl.n = (int)A.n;
    // (logistic: s_mblk is the g table; the term block sits behind it)
    sl.tab = LG ? (unsigned)(unsigned long long)(ldsc_t)s_mblk : 0u;
    sl.lds = (unsigned)(unsigned long long)(ldsc_t)(LG ? s_mblk + LG_LDS_DOUBLES + 2 : s_mblk);
    sl.NC = NC; sl.ncp = NCP; sl.p = p; sl.ic = ic; sl.nslots = A.sh_nslots; sl.lane0 = (int)blockIdx.x * LPW; sl.lcg = A.sh_lcg;
    if constexpr (LG) shard_long<FMCMC_FAM_LOGISTIC, 16>(sl);
    else if (p <= 16) shard_long<FMCMC_FAM_GAUSSIAN_LINREG, 16>(sl);
    else shard_long<FMCMC_FAM_GAUSSIAN_LINREG, 62>(sl);      // (wide linear models beyond the slices of the matrix-core form)
  } else if constexpr (FAM == FMCMC_FAM_LOGISTIC) {
    eval_sharded_logit_step<LPW>(A, s_mblk);
  } else {
  ShardCols sc;
  sc.xs = A.sh_xs + (long long)blockIdx.x * p * SH_MAXO;
  sc.ys = A.sh_ys + (long long)blockIdx.x * SH_MAXO;
  sc.th = A.sh_th; sc.part = A.sh_part; sc.n = A.n; sc.NC = NC; sc.p = p; sc.ic = ic; sc.nslots = A.sh_nslots;
  sc.lane0 = (int)blockIdx.x * LPW; sc.debug = A.debug;
  if (s_mblk) {
    ShardMfma sm;
    sm.th = A.sh_th; sm.part = A.sh_part; sm.NC = NC; sm.p = p; sm.ic = ic; sm.lane0 = (int)blockIdx.x * LPW; sm.tcount = 0;
    sm.ncp = NCP; sm.cstride = 1; sm.coff = 0; sm.thoff = 0; sm.tfirst = (int)(threadIdx.x >> 6); sm.tstep = NW;   // all chains, N-tiles round robin
    sm.lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const double*)s_mblk;
    const int KB = (p + 3) >> 2;
    if (A.sh_nmt == 1) shard_columns_mfma<LPW, 1>(sm);
    else if (A.sh_nmt == 2) shard_mfma_dispatch<LPW, 2>(sm, KB, 0);
    else if (A.sh_nmt == 4) shard_mfma_dispatch<LPW, 4>(sm, KB, 0);      // (slices of 41 .. 96 observations: 10,240 < n <= 24,576 at 256 workgroups)
    else if (A.sh_nmt == 5) shard_mfma_dispatch<LPW, 5>(sm, KB, 0);
    else if (A.sh_nmt == 6) shard_mfma_dispatch<LPW, 6>(sm, KB, 0);
    else shard_mfma_dispatch<LPW, 3>(sm, KB, A.sh_t10);
  } else
  if (A.sh_nslots * LPW <= SH_MAXO / 2) shard_columns<LPW, SH_MAXO / 2>(sc);   // half-empty slices: half the FMAs
  else shard_columns<LPW, SH_MAXO>(sc);
  }
  FMH_STAMP(stp, 5);
  if (!(A.debug & 32) && !(epoch & LOST)) { ok = shard_barrier(A.sh_bar, ++epoch, (A.debug & 512) != 0); if (!ok) epoch |= LOST; }
  FMH_STAMP(stp, 6);
  ok = !(epoch & LOST);
  // 3. thread = canonical lane: its partial of this workgroup's chains
#pragma unroll
  for (int c = 0; c < CW; c++)
    acc[c] = (c < ncw && !(A.debug & 128)) ? sh_load(A.sh_part + ((unsigned int)(cg0 + c) * (unsigned int)(NT + SH_PAD) + (unsigned int)tid)) : 1.0;
  return ok;
}

// ---- logistic family: sum_i g(|eta_i|) per canonical lane (twin of the oracle's logistic loop; R: vignettes/
// workflow-with-fmcmc.Rmd:35-41).  The linear part sum_i (y_i - 1/2) eta_i = sum_j b_j hs_j never enters a loop: hs (and the
// columns' largest |x|, for the range check below) come from logit_hs_kernel, once per launch.
template <int CW, bool LDSTAB>
__device__ __forceinline__ void logit_add_terms(const double (&es)[CW] /* 64 eta */, double (&acc)[CW], const double* s_tab) {
  double us[CW], gv[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) us[c] = __builtin_fabs(es[c]);
  logit_g_vec<CW, LDSTAB>(us, gv, s_tab);
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = acc[c] + gv[c];
}

// The evaluation of the logistic-only instantiations (PL covariates known at compile time, the CW (PL + 1) coefficients in
// SGPRs for the whole pass, table in LDS) as a REAL function: the sweep kernel calls it from two places (row 1 and the
// step loop), and inlined the copy inside the step loop shared its register allocation with everything the sweep keeps
// live across an evaluation.  Arguments travel in registers (16 dwords); everything uniform goes through v_readfirstlane.
struct LogitEval {
  const double* X;       // [p][n]
  const double* hs;      // [ic + p] data-only sums, then [p] largest |x| of every column (logit_hs_kernel)
  long long n;
  int ic, p;
  unsigned th0, thstride, ncw;   // LDS address of chain 0's SCALED coefficient vector (64 b), bytes between chains, chains of the workgroup
  unsigned tab;          // LDS address of the table (logistic-only instantiations)
  unsigned part;         // LDS address of s_part [NW][CW]
};
static_assert(sizeof(LogitEval) <= 64, "LogitEval must travel in registers (16 dwords)");
__device__ __forceinline__ unsigned logit_th_addr(const LogitEval& a, int c) {   // (a slot without a chain re-reads chain 0)
  const unsigned ncw = (unsigned)__builtin_amdgcn_readfirstlane((int)a.ncw);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)a.th0) + ((unsigned)c < ncw ? (unsigned)c : 0u) * (unsigned)__builtin_amdgcn_readfirstlane((int)a.thstride);
}
template <int CW, int PL>
__device__ __attribute__((noinline)) void logit_partials(LogitEval a) {
  typedef __attribute__((address_space(3))) const double* ldsc_t;
  typedef __attribute__((address_space(3))) double* ldsw_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ic = __builtin_amdgcn_readfirstlane(a.ic);
  const unsigned int nn = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned)a.n);
  const unsigned long long Xu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)a.X >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)a.X);
  const double* hs = (const double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)a.hs >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)a.hs));
  double b0[CW], bs[CW][PL > 0 ? PL : 1];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const ldsc_t th = (ldsc_t)(unsigned long long)logit_th_addr(a, c);
    b0[c] = ic ? sgpr_d(th[0]) : 0.0;
#pragma unroll
    for (int u = 0; u < PL; u++) bs[c][u] = sgpr_d(th[ic + u]);
  }
  // Range: |64 eta| <= |64 b0| + sum_u |64 b_u| max_i |x_iu| (+ a few ulps); below the table's 2400 rows NO observation needs the
  // "beyond the table" branch and the loop runs without it (uniform: the coefficients are).  Otherwise every element checks.
  bool fast = true;
#pragma unroll
  for (int c = 0; c < CW; c++) {
    double B = __builtin_fabs(b0[c]);
#pragma unroll
    for (int u = 0; u < PL; u++) B = fmh_fma(__builtin_fabs(bs[c][u]), sgpr_d(hs[ic + PL + u]), B);
    fast = fast && (B < (double)(FMH_LG_ROWS - 1));
  }
  fast = __builtin_amdgcn_ballot_w64(!fast) == 0ull;
  // global (not generic) pointers, scalar column bases and a 32-bit lane index: one global_load per value with no
  // 64-bit address arithmetic
  typedef const double __attribute__((address_space(1))) * gptr_t;
  typedef const char __attribute__((address_space(1))) * gcptr_t;
  gptr_t colp[PL > 0 ? PL : 1];
#pragma unroll
  for (int u = 0; u < PL; u++) colp[u] = (gptr_t)(Xu + 8ull * (unsigned long long)u * nn);
  // (the lane's offset in BYTES as a 32-bit value -- n < 2^28 -- so that every load is `scalar base + 32-bit VGPR offset`)
  auto ldg = [](gptr_t base, unsigned int boff) -> double { return *(gptr_t)((gcptr_t)base + boff); };
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  // ---- the observation loop.  Per element EXACTLY the operations of fmh_logit_g_scaled (the oracle's bits): 64 eta as the fma
  // chain over the scaled coefficients, s = fract, j = trunc, three 16-byte table reads at one address + immediate offsets,
  // five fmas, one add.  Two operand sets used alternately: an observation's columns are re-loaded for the observation TWO
  // passes ahead the moment its eta is formed; the trip count is uniform (scalar branch), the last pass runs once more under
  // the exec mask, the prefetch index is clamped with one v_min_u32.  The intercepts sit in VGPRs: as SGPRs each cost a
  // v_mov_b64 per observation (an fma takes one scalar operand).
  typedef __attribute__((address_space(3))) const char* ldsb_t;
  typedef __attribute__((address_space(3))) const lg_v2d* lds2_t;
  auto vconst = [](double c) -> double { asm volatile("" : "+v"(c)); return c; };
  const unsigned int tabaddr = (unsigned int)__builtin_amdgcn_readfirstlane((int)a.tab);
  const unsigned int blast = 8u * (nn - 1u);
  constexpr int PLX = PL > 0 ? PL : 1;
  double xb0[PLX], xb1[PLX];
  double b0v[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) b0v[c] = vconst(b0[c]);
  unsigned int boff = 8u * (unsigned int)tid;          // byte offset of the observation the NEXT reload is for, minus one pass
  {
    const unsigned int b = boff < blast ? boff : blast;
#pragma unroll
    for (int u = 0; u < PL; u++) xb0[u] = ldg(colp[u], b);
    boff += 8u * NT;
    const unsigned int b1 = boff < blast ? boff : blast;
#pragma unroll
    for (int u = 0; u < PL; u++) xb1[u] = ldg(colp[u], b1);
  }
  auto one_observation = [&](double (&xb)[PLX], auto checked) {
    constexpr bool CHECKED = decltype(checked)::value;
    double es[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) es[c] = b0v[c];
#pragma unroll
    for (int u = 0; u < PL; u++) {
#pragma unroll
      for (int c = 0; c < CW; c++) es[c] = fmh_fma(xb[u], bs[c][u], es[c]);
    }
    boff += 8u * NT;
    {
      const unsigned int b = boff < blast ? boff : blast;         // clamped: the last prefetches re-read the last observation
#pragma unroll
      for (int u = 0; u < PL; u++) xb[u] = ldg(colp[u], b);
    }
    double sv[CW];
    lg_v2d p0[CW], p1[CW], p2[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) {
      unsigned int j;     // (|es| as the source modifier of both consumers, as in logit_shard)
      asm("v_fract_f64_e64 %0, |%1|" : "=v"(sv[c]) : "v"(es[c]));
      asm("v_cvt_u32_f64_e64 %0, |%1|" : "=v"(j) : "v"(es[c]));
      if (CHECKED) j = (j < (unsigned)FMH_LG_ROWS) ? j : (unsigned)(FMH_LG_ROWS - 1);
      const ldsb_t row = (ldsb_t)(unsigned long long)(tabaddr + 48u * j);
      p0[c] = *(lds2_t)(row);
      p1[c] = *(lds2_t)(row + 16);
      p2[c] = *(lds2_t)(row + 32);
    }
#pragma unroll
    for (int c = 0; c < CW; c++) {
      double q = fmh_fma(sv[c], p2[c].y, p2[c].x);
      q = fmh_fma(sv[c], q, p1[c].y);
      q = fmh_fma(sv[c], q, p1[c].x);
      q = fmh_fma(sv[c], q, p0[c].y);
      q = fmh_fma(sv[c], q, p0[c].x);
      if (CHECKED) { const double ue = __builtin_fabs(es[c]); q = (ue < (double)FMH_LG_ROWS) ? q : ue * FMH_LG_HALF_INV_SCALE; }
      acc[c] = acc[c] + q;
    }
  };
  const unsigned int T = (nn + NT - 1u) / NT;        // uniform
  const bool last_valid = (unsigned int)tid + NT * (T - 1u) < nn;
  auto all_observations = [&](auto checked) {
    unsigned int it = 0;
    for (; it + 2u < T; it += 2u) { one_observation(xb0, checked); one_observation(xb1, checked); }
    if (T - it == 2u) {
      one_observation(xb0, checked);
      if (last_valid) one_observation(xb1, checked);
    } else if (T - it == 1u) {
      if (last_valid) one_observation(xb0, checked);
    }
  };
  if (fast) all_observations(std::false_type{});
  else all_observations(std::true_type{});
  const ldsw_t s_part = (ldsw_t)(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)a.part);
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const double v = wave_xor_sum(acc[c]);
    if (lane == 0) s_part[wave * CW + c] = v;
  }
}

// Every other logistic shape (any p, any n; all-family kernels: table in global memory), a real function for the same reason.
template <int CW, bool LDSTAB>
__device__ __attribute__((noinline)) void logit_partials_any(LogitEval a) {
  typedef __attribute__((address_space(3))) const double* ldsc_t;
  typedef __attribute__((address_space(3))) double* ldsw_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ic = __builtin_amdgcn_readfirstlane(a.ic), p = __builtin_amdgcn_readfirstlane(a.p);
  const long long n = (long long)rfl_u64((unsigned long long)a.n);
  const double* X = (const double*)rfl_u64((unsigned long long)a.X);
  const double* s_tab = LDSTAB ? (const double*)(ldsc_t)(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)a.tab) : nullptr;
  ldsc_t th[CW];                       // the SCALED coefficient vectors (64 b)
#pragma unroll
  for (int c = 0; c < CW; c++) th[c] = (ldsc_t)(unsigned long long)logit_th_addr(a, c);
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  // The data comes from L2 (~1 us under load) and only two waves share a SIMD: a load-use chain per column made this
  // loop latency-bound (6 dependent round trips per observation).  All columns of an observation are fetched as one
  // batch, and the batch of the NEXT observation is in flight while the terms of the current one are formed.
  constexpr int JB = 8;
  if (p <= JB) {
    double xb[JB];
    long long i = tid;
    if (i < n) {
#pragma unroll
      for (int u = 0; u < JB; u++) xb[u] = (u < p) ? X[(long long)u * n + i] : 0.0;
    }
    for (; i < n; i += NT) {
      double es[CW];
#pragma unroll
      for (int c = 0; c < CW; c++) es[c] = ic ? th[c][0] : 0.0;
#pragma unroll
      for (int u = 0; u < JB; u++)
        if (u < p) {
#pragma unroll
          for (int c = 0; c < CW; c++) es[c] = fmh_fma(xb[u], th[c][ic + u], es[c]);
        }
      const long long inx = (i + NT < n) ? i + NT : i;   // clamped: the last prefetch re-reads this observation
#pragma unroll
      for (int u = 0; u < JB; u++) xb[u] = (u < p) ? X[(long long)u * n + inx] : 0.0;
      logit_add_terms<CW, LDSTAB>(es, acc, s_tab);
    }
  } else {
    for (long long i = tid; i < n; i += NT) {
      double es[CW];
#pragma unroll
      for (int c = 0; c < CW; c++) es[c] = ic ? th[c][0] : 0.0;
      int j = 0;
      for (; j + JB <= p; j += JB) {
        double xb[JB];
#pragma unroll
        for (int u = 0; u < JB; u++) xb[u] = X[(long long)(j + u) * n + i];
#pragma unroll
        for (int u = 0; u < JB; u++) {
#pragma unroll
          for (int c = 0; c < CW; c++) es[c] = fmh_fma(xb[u], th[c][ic + j + u], es[c]);
        }
      }
      for (; j < p; j++) {
        const double x = X[(long long)j * n + i];
#pragma unroll
        for (int c = 0; c < CW; c++) es[c] = fmh_fma(x, th[c][ic + j], es[c]);
      }
      logit_add_terms<CW, LDSTAB>(es, acc, s_tab);
    }
  }
  const ldsw_t s_part = (ldsw_t)(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)a.part);
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const double v = wave_xor_sum(acc[c]);
    if (lane == 0) s_part[wave * CW + c] = v;
  }
}

// ---- observation-sharded logistic evaluation (config C5; the decomposition of the wide linear models above) ---------------
// Chain-sharded, every workgroup streams the whole data set from L2 once per step (4 MB at C5: the per-CU L2 rate alone is
// ~62 us per step, tools/probe_logit_grid.hip) and its lanes -- observations -- scatter over the g table (~2.9-way bank
// conflicts on every lookup).  Here workgroup b of 256 owns the canonical lanes 2b, 2b + 1 for ALL chains: a constant slice of
// 2 x ceil(n / 512) observations whose covariates arrive as SCALAR operands (s_load from a compact copy, logit_build_slices),
// thread = chain (NCH chains per thread side by side), the chain's scaled coefficients in its lane's VGPRs.  The lanes of a
// wave then look up g for the SAME observation and neighbouring chains: equal or neighbouring rows, no conflicts once the
// chains have found the posterior.  Per observation and chain: 5 + 2 + 1 + 5 + 1 vector instructions and three ds_read_b128
// (every instruction of any kind costs a SIMD ~4.5 cycles at two waves: tools/probe_dp_ops.hip), against 39 in the round-3 loop.
// A pass = one slot = the two observations of the slice's two canonical lanes; the loop is software-pipelined BY HAND: a pass
// issues the lookups of pass p + 1 (one burst of ds_read_b128) and then runs the polynomials of pass p, whose coefficients
// were read a pass ago, so the LDS round trip hides under a pass of arithmetic.  ONE wait per pass, at its top: lgkmcnt(0) --
// scalar loads share the counter with LDS reads and return out of order, a counted wait is not available -- and the scalar
// loads of pass p + 2 go out right behind it (tools/probe_logit_shard.hip: 98 us per evaluation left to the compiler's
// schedule, 76 with the phases fenced, 70 pipelined).
struct LogitShard {
  const double* xs;      // this workgroup's slice [nslots][2][PL] (0 beyond n)
  const double* th;      // [k][ncp] proposals of all chains (unscaled)
  double* part;          // [NC][NT + SH_PAD] lane partials
  const double* hs;      // logit_hs_kernel's block: [ic + p] sums, [p] largest |x| per column
  unsigned tab;          // LDS address of the table
  int NC, ncp, ic, lane0;
  int nv0, nv1;          // slots of canonical lane lane0 / lane0 + 1 that hold an observation (nv1 <= nv0 <= nslots)
  int turn;              // thousandths of its passes for which the younger wave of a SIMD holds the issue priority
};
static_assert(sizeof(LogitShard) == 64, "LogitShard must stay at 16 dwords");
// OB = 2: a pass is one slot, the two observations of the slice's two canonical lanes (p <= 8); OB = 1: a pass is ONE observation,
// even passes the lower lane's, odd passes the upper lane's -- the covariates of two passes then fit the scalar registers up to
// p = 16 (round 4: tools/dispatch_audit.py found logistic models with more than 8 covariates on the run-time chain-sharded loop,
// 5x the time per step of p = 8).  Either way a canonical lane's terms are added in slot order: the same bits.
template <int PL, int NCH, int OB = 2>
__device__ __attribute__((noinline)) void logit_shard(LogitShard c) {
  static_assert(OB == 1 || OB == 2, "one or two observations per pass");
  typedef const double __attribute__((address_space(4))) * cptr_t;
  typedef __attribute__((address_space(3))) const char* ldsb_t;
  typedef __attribute__((address_space(3))) const lg_v2d* lds2_t;
  const int tid = threadIdx.x;
  const int NC = rfl_i(c.NC), NCP = rfl_i(c.ncp), ic = rfl_i(c.ic), lane0 = rfl_i(c.lane0), nv0 = rfl_i(c.nv0), nv1 = rfl_i(c.nv1);
  const unsigned tabaddr = (unsigned)rfl_i((int)c.tab);
  const cptr_t slice = (cptr_t)rfl_u64((unsigned long long)c.xs);
  const double* thg = (const double*)rfl_u64((unsigned long long)c.th);
  const double* hs = (const double*)rfl_u64((unsigned long long)c.hs);
  double* part = (double*)rfl_u64((unsigned long long)c.part);
  double cmax[PL];
#pragma unroll
  for (int u = 0; u < PL; u++) cmax[u] = sgpr_d(hs[ic + PL + u]);
  const int npass = nv1 < nv0 ? nv1 : nv0;       // slots in which BOTH lanes hold an observation
  for (int cb = 0; cb < NC; cb += NT * NCH) {
    double b0[NCH], bs[NCH][PL], a0[NCH], a1[NCH];
    bool fast = true;
#pragma unroll
    for (int h = 0; h < NCH; h++) {
      const int chain = cb + h * NT + tid;
      const unsigned int chc = (unsigned int)(chain < NC ? chain : 0);
      b0[h] = ic ? sh_load(thg + chc) * FMH_LG_SCALE : 0.0;
      double B = __builtin_fabs(b0[h]);
#pragma unroll
      for (int u = 0; u < PL; u++) {
        bs[h][u] = sh_load(thg + ((unsigned int)((ic + u) * NCP) + chc)) * FMH_LG_SCALE;
        B = fmh_fma(__builtin_fabs(bs[h][u]), cmax[u], B);
      }
      fast = fast && (B < (double)(FMH_LG_ROWS - 1));   // no observation of this chain leaves the table (NaN: checked form)
      a0[h] = 0.0; a1[h] = 0.0;
    }
    fast = __builtin_amdgcn_ballot_w64(!fast) == 0ull;
    // one observation (pass, q) for chain slot h, every element checked: the tail slot and the waves with a chain off the table
    auto checked_term = [&](int pass, int q, int h) -> double {
      double es = b0[h];
#pragma unroll
      for (int u = 0; u < PL; u++) es = fmh_fma(slice[(pass * 2 + q) * PL + u], bs[h][u], es);
      const double us1[1] = {__builtin_fabs(es)};
      double g1[1];
      logit_g_vec<1, true>(us1, g1, (const double*)(__attribute__((address_space(3))) const double*)(unsigned long long)tabaddr);
      return g1[0];
    };
    if (fast && npass > 0) {
      double xa[OB][PL], xb[OB][PL], sva[OB][NCH], svb[OB][NCH];
      lg_v2d pra[OB][NCH][3], prb[OB][NCH][3];
      // (a pass's covariates by a RUNNING pointer + immediate offsets: indexed by the pass number every load group cost a sign
      //  extension, a shift and a 64-bit add of scalar instructions, and every instruction of any kind is ~4.5 cycles of the SIMD;
      //  beyond the end the loads read the next slice or the padding behind the last one -- logit_build_slices -- never accumulated)
      auto sload = [&](double (&x)[OB][PL], cptr_t base) {
#pragma unroll
        for (int q = 0; q < OB; q++)
#pragma unroll
          for (int u = 0; u < PL; u++) x[q][u] = base[q * PL + u];
      };
      auto front = [&](const double (&x)[OB][PL], double (&sv)[OB][NCH], lg_v2d (&pr)[OB][NCH][3]) {   // 64 eta, reduction, lookups
        unsigned ad[OB][NCH];
#pragma unroll
        for (int q = 0; q < OB; q++)
#pragma unroll
          for (int h = 0; h < NCH; h++) {
            double es = b0[h];
#pragma unroll
            for (int u = 0; u < PL; u++) es = fmh_fma(x[q][u], bs[h][u], es);
            // (|es| as the source modifier of BOTH consumers: left to itself the compiler cleared the sign bit with a v_and_b32 of its own
            //  in half of the cases -- 0.5 of 14.5 vector instructions per observation and chain)
            unsigned rowi;
            asm("v_fract_f64_e64 %0, |%1|" : "=v"(sv[q][h]) : "v"(es));
            asm("v_cvt_u32_f64_e64 %0, |%1|" : "=v"(rowi) : "v"(es));
            ad[q][h] = __umul24(rowi, 48u) + tabaddr;   // (v_mad_u32_u24: one instruction; the plain product became a 64-bit mad)
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < OB; q++)
#pragma unroll
          for (int h = 0; h < NCH; h++)
#pragma unroll
            for (int k = 0; k < 3; k++) pr[q][h][k] = *(lds2_t)((ldsb_t)(unsigned long long)ad[q][h] + 16 * k);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto back = [&](const double (&sv)[OB][NCH], const lg_v2d (&pr)[OB][NCH][3], const int upper /* OB == 1: the pass's lane */) {   // polynomials
#pragma unroll
        for (int q = 0; q < OB; q++)
#pragma unroll
          for (int h = 0; h < NCH; h++) {
            double v = fmh_fma(sv[q][h], pr[q][h][2].y, pr[q][h][2].x);
            v = fmh_fma(sv[q][h], v, pr[q][h][1].y);
            v = fmh_fma(sv[q][h], v, pr[q][h][1].x);
            v = fmh_fma(sv[q][h], v, pr[q][h][0].y);
            v = fmh_fma(sv[q][h], v, pr[q][h][0].x);
            if (OB == 2 ? q : upper) a1[h] = a1[h] + v; else a0[h] = a0[h] + v;
          }
        __builtin_amdgcn_sched_barrier(0);
      };
      cptr_t xp = slice;                          // covariates of pass ps
      const int nps = npass * (2 / OB);           // passes in which the pipelined loop runs (OB == 1: two per slot)
      sload(xa, xp);
      sload(xb, xp + OB * PL);
      front(xa, sva, pra);                        // pass 0's lookups in flight
      // ISSUE PRIORITY IN TWO TURNS (round 5).  The arbiter of a SIMD serves its OLDEST ready wave first: of the two waves of a SIMD
      // (w and w + 4) the older one ran this loop at the pace of a wave that has the SIMD to itself and was done after 38.9 us, the
      // younger one filled the gaps and then ran its last 17 us ALONE (stamps of all eight waves, tools/stamp_c5.py: 38.9 / 55.7 us)
      // -- and one wave alone issues at ~0.64 of the rate of two.  Now the YOUNGER wave holds priority 1 for its first `nturn`
      // passes and drops to 0 for the rest: first it is the one that runs ahead, then the older one, and the two finish together.
      // nturn REGULATES ITSELF: both waves leave their finishing time (s_memtime) in LDS, and at its next call the younger one
      // moves its turn by (its time - the partner's + bias) / 512 cycles, at most 8 passes -- the balance point depends on p, on
      // the chains per thread and on the clock, and a fixed fraction that was right for config C5 (0.72) was wrong by 2 % at 0.675.
      // Measured at C5 (tools/bench_c5_turn.sh, us per step): no turn 61.2; fixed 0.64 / 0.72: 59.9 / 58.9; regulated with the
      // younger wave meant to finish 0 / 2048 / 4096 / 6144 / 8192 cycles BEFORE its partner: 59.8 / 59.2 / 58.9 / 59.1 / 59.3.
      // c.turn = thousandths to start from + 10000 (stay there) + 100000 x (bias / 256 cycles).
      // Timing only: no result depends on it.  (Priority alternating pass by pass was measured too: 68.3 us per step against 61.1;
      // a compare and a branch around s_setprio inside the loop split its basic block: two copies of the loop instead.)
      typedef __attribute__((address_space(3))) unsigned* ldsu_t;
      const int wv = rfl_i(tid >> 6), simd = wv & 3;
      const bool young = wv >= NW / 2;
      const ldsu_t ctl = (ldsu_t)(unsigned long long)(tabaddr + (unsigned)(LG_LDS_DOUBLES * 8));   // [4] nturn | [4] end of the older | [4] of the younger wave
      const int tset = rfl_i(c.turn);                      // thousandths of the passes to start from (+ 10000: and to stay at)
      int nturn = 0;
      if (young) {
        int cur = rfl_i((int)ctl[simd]);
        if (cur < 0 || cur > nps) cur = (int)(((long long)nps * (tset % 10000)) / 1000);
        else if ((tset % 100000) < 10000) {
          const int diff = rfl_i((int)(ctl[8 + simd] - ctl[4 + simd])) + 256 * (tset / 100000);      // > 0: this wave was later than it is meant to be
          int step = diff / 512;
          step = step > 8 ? 8 : (step < -8 ? -8 : step);
          cur += step;
          cur = cur < 0 ? 0 : (cur > nps ? nps : cur);
        }
        if ((tid & 63) == 0) ctl[simd] = (unsigned)cur;
        nturn = cur & ~1;
      }
      auto passes = [&](const int from, const int to) {
        for (int ps = from; ps < to; ps += 2) {
          __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): pass ps's coefficients, pass ps + 1's covariates
          sload(xa, xp + 2 * OB * PL);
          front(xb, svb, prb);                      // pass ps + 1
          back(sva, pra, 0);                        // pass ps
          __builtin_amdgcn_s_waitcnt(0xC07F);
          sload(xb, xp + 3 * OB * PL);
          front(xa, sva, pra);                      // pass ps + 2
          if (ps + 1 < nps) back(svb, prb, 1);      // pass ps + 1
          xp += 2 * OB * PL;
        }
      };
      if (nturn > 0) {
        __builtin_amdgcn_s_setprio(1);
        passes(0, nturn);
        __builtin_amdgcn_s_setprio(0);
      }
      passes(nturn, nps);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if ((tid & 63) == 0) ctl[(young ? 8 : 4) + simd] = cycle_stamp32();
    } else {
      for (int ps = 0; ps < npass; ps++) {
#pragma unroll
        for (int h = 0; h < NCH; h++) { a0[h] = a0[h] + checked_term(ps, 0, h); a1[h] = a1[h] + checked_term(ps, 1, h); }
      }
    }
    for (int ps = npass; ps < nv0; ps++) {        // the last slot where only the lower lane holds an observation
#pragma unroll
      for (int h = 0; h < NCH; h++) a0[h] = a0[h] + checked_term(ps, 0, h);
    }
#pragma unroll
    for (int h = 0; h < NCH; h++) {
      const int chain = cb + h * NT + tid;
      if (chain < NC) sh_store2(&part[(long long)chain * (NT + SH_PAD) + lane0], a0[h], a1[h]);
    }
  }
}

template <int LPW>
__device__ __forceinline__ void eval_sharded_logit_step(const SweepArgs& A, const double* s_tab) {
  static_assert(LPW == 2, "the observation-sharded logistic evaluation is written for 256 workgroups of two canonical lanes");
  typedef __attribute__((address_space(3))) const double* ldsc_t;
  LogitShard ls;
  const int nslots = A.sh_nslots, lane0 = (int)blockIdx.x * LPW;
  ls.xs = A.sh_xs + (long long)blockIdx.x * nslots * 2 * A.p;
  ls.th = A.sh_th; ls.part = A.sh_part; ls.hs = A.lg_hs;
  ls.tab = (unsigned)(unsigned long long)(ldsc_t)s_tab;
  ls.NC = (int)A.nchains; ls.ncp = (int)A.nchains + SH_PAD; ls.ic = A.intercept; ls.lane0 = lane0;
  // slots of lane l that hold an observation: i = 512 slot + l < n
  auto nvalid = [&](int l) -> int { const long long v = (A.n - l + NT - 1) / NT; return (int)(v < 0 ? 0 : (v > nslots ? nslots : v)); };
  ls.nv0 = nvalid(lane0); ls.nv1 = nvalid(lane0 + 1); ls.turn = A.sh_t10;   // (sh_t10: this family's use of the field)
  const bool two = A.nchains > NT;     // (uniform) more than 512 chains in the launch: two chains per thread side by side
  switch (A.p) {
#define LG_CASE(P_) case P_: if (two) logit_shard<P_, 2>(ls); else logit_shard<P_, 1>(ls); break;
    LG_CASE(1) LG_CASE(2) LG_CASE(3) LG_CASE(4) LG_CASE(5) LG_CASE(6) LG_CASE(7) LG_CASE(8)
#undef LG_CASE
#define LG_CASE(P_) case P_: if (two) logit_shard<P_, 2, 1>(ls); else logit_shard<P_, 1, 1>(ls); break;
    LG_CASE(9) LG_CASE(10) LG_CASE(11) LG_CASE(12) LG_CASE(13) LG_CASE(14) LG_CASE(15) LG_CASE(16)
#undef LG_CASE
    default: break;                    // (the host takes this form for 1 <= p <= 16 only)
  }
}

// ---- workgroup-collective log-posterior partial sums (streamed variant) ------------------
// Every thread accumulates its canonical lane for all CW chains, then the wavefront butterfly
// (levels 1..32) runs and lane 0 of each wavefront publishes its partial to s_part[w][c].
// FAM > 0 compiles one family in (leaner kernels for the logistic model), FAM == 0 keeps all behind A.family.
template <int CW, int FAM = 0, int SHL = 0 /* > 0: observation-sharded evaluation, SHL canonical lanes per workgroup */>
__device__ __forceinline__ void eval_partials(const SweepArgs& A, double* const* th /*[CW] -> theta in LDS*/,
                                              double* s_part, const double* s_sptab = nullptr /* LDS: g table (logistic) / MFMA slice block (sharded linreg) */,
                                              unsigned* sh_epoch = nullptr /* barrier epoch of the sharded evaluation */, Stamps* stp = nullptr,
                                              double* s_lgb = nullptr /* LDS [CW][k]: scaled coefficient copies (logistic) */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long n = A.n;
  const int family = FAM ? FAM : A.family;
  const int p = (family == FMCMC_FAM_IID_NORMAL) ? 0 : A.p;
  const int ic = (family == FMCMC_FAM_IID_NORMAL) ? 1 : A.intercept;
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  if (family == FMCMC_FAM_LOGISTIC) {
    typedef __attribute__((address_space(3))) const double* ldsc_t;
    if constexpr (FAM == FMCMC_FAM_LOGISTIC && SHL > 0) {
      // observation-sharded: this thread's canonical lane of the workgroup's chains comes back from the grid (eval_sharded),
      // then the wave butterfly as everywhere
      const bool ok = eval_sharded<CW, SHL, FMCMC_FAM_LOGISTIC>(A, th, acc, *sh_epoch, s_sptab, stp);
      if (!ok && tid == 0 && (long long)blockIdx.x * CW < A.nchains) A.status[(long long)blockIdx.x * CW] = FMCMC_CHAIN_SYNC_TIMEOUT;
#pragma unroll
      for (int c = 0; c < CW; c++) {
        const double v = wave_xor_sum(acc[c]);
        if (lane == 0) s_part[wave * CW + c] = v;
      }
      return;
    }
    // Real functions with their own register allocation (logit_partials / logit_partials_any above); they also run the wave
    // butterfly and publish s_part.  P known at compile time (the logistic-only instantiations, CW (P + 1) <= 28
    // coefficients in SGPRs): straight-line loop body; with a run-time p every `if (u < p)` was a basic block of its own.
    // The loops form 64 eta from SCALED coefficients (the table's argument, include/fmh_detmath.h): copies of the workgroup's
    // coefficient vectors times 64 (exact), refreshed here for every evaluation.
    const int nb = ic + p;
    {
      int nlive = 1;
#pragma unroll
      for (int c = 1; c < CW; c++) if (th[c] != th[0]) nlive = c + 1;
      lds_barrier();                                    // (the previous evaluation's readers are done with the copies)
      for (int idx = tid; idx < nlive * nb; idx += NT) {
        const int c = idx / nb, j = idx - c * nb;
        s_lgb[c * A.k + j] = th[c][j] * FMH_LG_SCALE;
      }
      lds_barrier();
    }
    LogitEval le;
    le.X = A.X; le.hs = A.lg_hs; le.n = n; le.ic = ic; le.p = p;
    le.th0 = (unsigned)(unsigned long long)(ldsc_t)s_lgb;
    // (the caller's layout: th[c] = th[0] + c * stride for the chains the workgroup holds, th[0] again for empty slots)
    le.thstride = (unsigned)(A.k * sizeof(double)); le.ncw = 1u;
#pragma unroll
    for (int c = 1; c < CW; c++) if (th[c] != th[0]) le.ncw = (unsigned)c + 1u;
    le.tab = (unsigned)(unsigned long long)(ldsc_t)s_sptab;
    le.part = (unsigned)(unsigned long long)(ldsc_t)s_part;
    constexpr int PMAX = (FAM == FMCMC_FAM_LOGISTIC) ? (28 / CW - 1 > 8 ? 8 : 28 / CW - 1) : -1;
    if constexpr (FAM == FMCMC_FAM_LOGISTIC) {
      if (p <= PMAX && n < (1ll << 28)) {
        switch (p) {
          case 0: logit_partials<CW, 0>(le); break;
          case 1: logit_partials<CW, (PMAX >= 1 ? 1 : 0)>(le); break;
          case 2: logit_partials<CW, (PMAX >= 2 ? 2 : 0)>(le); break;
          case 3: logit_partials<CW, (PMAX >= 3 ? 3 : 0)>(le); break;
          case 4: logit_partials<CW, (PMAX >= 4 ? 4 : 0)>(le); break;
          case 5: logit_partials<CW, (PMAX >= 5 ? 5 : 0)>(le); break;
          case 6: logit_partials<CW, (PMAX >= 6 ? 6 : 0)>(le); break;
          case 7: logit_partials<CW, (PMAX >= 7 ? 7 : 0)>(le); break;
          default: logit_partials<CW, (PMAX >= 8 ? 8 : 0)>(le); break;
        }
        return;
      }
    }
    logit_partials_any<CW, FAM == FMCMC_FAM_LOGISTIC>(le);
    return;
  } else if constexpr (FAM == FMCMC_FAM_GAUSSIAN_LINREG && CW <= 2 && SHL > 0) {
    const bool ok = eval_sharded<CW, SHL>(A, th, acc, *sh_epoch, s_sptab, stp);
    if (!ok && tid == 0 && (long long)blockIdx.x * CW < A.nchains)      // (a workgroup without chains has no status slot)
      A.status[(long long)blockIdx.x * CW] = FMCMC_CHAIN_SYNC_TIMEOUT;
  } else {
    // Memory-level parallelism: the data comes from L2 (latency ~1 us under load), so every thread keeps a
    // batch of JB independent column loads in flight before the FMAs that consume them; a dependent
    // load-use chain per (observation, column) left < 16 KB in flight per CU (10x below the L2 rate at k = 50).
    // Measured (tools/bench_cw.py, k = 50, 512 chains): 2 chains per workgroup is the optimum (59 us per step; 1: 114,
    // 4: 70, 8: 120), i.e. the loop is bound by load latency + FMA issue per CU (~65 GB/s per CU of the 154 GB/s L1 fill
    // rate), not by aggregate L2 bandwidth; a second batch in flight (double-buffered xb) spills in this all-kinds kernel
    // and is 12 % slower.
    // Wide models (p >= 16, e.g. config C4: k = 50): every thread works on OB of its observations at once, so that every
    // coefficient read from LDS feeds OB observations and OB x 8 column loads are in flight per thread.  Same arithmetic
    // per observation, and the lane still accumulates its observations in index order.  Worth 2 % only, and no other
    // form of this loop is faster (column batches outside / observations inside with the coefficients in SGPRs and 40
    // buffer loads in flight per thread: same time): at k = 50 a CU reads X at 65-67 GB/s, the rate at which one CU
    // gets data out of its XCD's L2 (MI355X_MICROARCH.md: 66-73 GB/s per CU) -- DESIGN.md, C4.
    constexpr int JB = 8;
#ifndef FMCMC_OBS_BLOCK
#define FMCMC_OBS_BLOCK 2
#endif
    // (only the one-family, one-kernel instantiations have the registers for it; 4 chains x 4 observations never fit)
    constexpr int OB = (CW <= 2 && FAM == FMCMC_FAM_GAUSSIAN_LINREG) ? FMCMC_OBS_BLOCK : 1;
    if (OB > 1 && p >= 2 * JB) {
      for (long long i0 = tid; i0 < n; i0 += (long long)NT * OB) {
        long long ii[OB];
        double mu[OB][CW];
#pragma unroll
        for (int o = 0; o < OB; o++) {
          const long long i = i0 + (long long)NT * o;
          ii[o] = (i < n) ? i : i0;            // clamped: a padded slot re-reads observation i0 and is discarded below
#pragma unroll
          for (int c = 0; c < CW; c++) mu[o][c] = ic ? th[c][0] : 0.0;
        }
        int j = 0;
        for (; j + JB <= p; j += JB) {
          double xb[OB][JB];
#pragma unroll
          for (int u = 0; u < JB; u++) {
#pragma unroll
            for (int o = 0; o < OB; o++) xb[o][u] = A.X[(long long)(j + u) * n + ii[o]];
          }
#pragma unroll
          for (int u = 0; u < JB; u++) {
#pragma unroll
            for (int c = 0; c < CW; c++) {
              const double b = th[c][ic + j + u];
#pragma unroll
              for (int o = 0; o < OB; o++) mu[o][c] = fmh_fma(xb[o][u], b, mu[o][c]);
            }
          }
        }
        for (; j < p; j++) {
#pragma unroll
          for (int c = 0; c < CW; c++) {
            const double b = th[c][ic + j];
#pragma unroll
            for (int o = 0; o < OB; o++) mu[o][c] = fmh_fma(A.X[(long long)j * n + ii[o]], b, mu[o][c]);
          }
        }
#pragma unroll
        for (int o = 0; o < OB; o++) {
          const double yv = A.y[ii[o]];
          const bool valid = (i0 + (long long)NT * o) < n;
#pragma unroll
          for (int c = 0; c < CW; c++) {
            const double r = valid ? yv - mu[o][c] : 0.0;       // fma(0, 0, acc) == acc exactly
            acc[c] = fmh_fma(r, r, acc[c]);
          }
        }
      }
    } else
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
template <int FAM = 0>
__device__ __forceinline__ double finish_logpost(const SweepArgs& A, const double* th, double tot, const double* hs = nullptr /* LDS copy of A.lg_hs */) {
  double f;
  const int family = FAM ? FAM : A.family;
  if (family == FMCMC_FAM_LOGISTIC) {
    // logl = sum_j b_j hs_j - sum_i g(|eta_i|) (include/fmh_detmath.h, fmh_logit_g; the oracle's logpost_canon)
    double lin = 0.0;
    if (!hs) hs = A.lg_hs;
    for (int j = 0; j < A.intercept + A.p; j++) lin = fmh_fma(th[j], hs[j], lin);
    f = lin - tot;
    if (A.prior_div != 0.0) {
      double ss = 0.0;
      const int nb = A.intercept + A.p;
      for (int j = 0; j < nb; j++) ss = fmh_fma(th[j], th[j], ss);
      f = f - ss / A.prior_div;
    }
  } else {
    const int pp = (family == FMCMC_FAM_IID_NORMAL) ? 0 : A.p;
    const int ic = (family == FMCMC_FAM_IID_NORMAL) ? 1 : A.intercept;
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
