// mh_engine.hip — gfx950 (MI355X) many-chain Metropolis-Hastings engine: the C-ABI (include/fmcmc_amd.h), validation,
// kernel selection and launches.  The sweep kernels are instantiated in the k_*.hip translation units (compiled in parallel,
// fmcmc_amd/build.py) and reached through the look-ups of mh_kernels.hpp; their source is in the headers:
//   mh_common.hpp  shared device helpers      mh_streamed.hpp  general kernel (all families / kernels / schemes)
//   mh_rng.hpp     RNG stream kernel          mh_mfma.hpp      fp64-MFMA kernel, owner waves (headline)
//   mh_spec.hpp    wave-specialised kernel (kernel_adapt / kernel_ram; the latency form for few chains)
//   mh_wide2.hpp   wide models: observation-sharded dataflow kernel (owner / evaluator waves, two chain groups)
//   mh_mfma_ad.hpp adaptive owners on the MFMA evaluation   mh_bigk.hpp  more than 64 parameters
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
#define FMH_WITH_RNG_FILL
#include "mh_tu.hpp"
#include <vector>
// (host-side shape helpers of the kernel families -- LDS sizes, slot counts; their kernels are instantiated in the k_*.hip files)
#include "mh_streamed.hpp"
#include "mh_mfma.hpp"
#include "mh_spec.hpp"
#include "mh_lat.hpp"
#include "mh_wide2.hpp"
#include "mh_mfma_ad.hpp"
#include "mh_bigk.hpp"

namespace {

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
    case 9: r = fmh_logit_g(v); break;   // g(|v|), the per-observation term of the logistic family
    case 10: r = fmh_unif(seed, (unsigned)(i & 0xffff), (unsigned)(i >> 16), (unsigned)(i % 7)); break;
    case 12: r = fmh_tan_0_halfpi(v); break;
    default: r = fmh_nan();
  }
  out[i] = r;
}

// the observation slots beyond the operand registers of mh_sweep_mfma<.., EXT>, in operand order: for wave w, streamed slot e
// (observation slot ns_res + e), group q, lane l, lane-group value g: column 4 q + l / 16 of [x_1 .. x_p, y, 0 ..] for
// observation i = 64 w + cl_a(l % 16) + g + 512 (ns_res + e); 0 beyond n.  out[((((w next + e) ng + q) 64 + l) 4 + g]
__global__ void mfma_build_stream(const double* X, const double* y, long long n, int p, int ng, int ns_res, int next, double* out) {
  const long long total = (long long)NW * next * ng * 64 * 4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int q = (int)(r % ng); r /= ng;
    const int e = (int)(r % next), w = (int)(r / next);
    const int f = 4 * q + (l >> 4), o16 = l & 15;
    const long long i = (long long)(64 * w + 16 * (o16 & 3) + 4 * (o16 >> 2) + g) + (long long)NT * (ns_res + e);
    double a = 0.0;
    if (i < n) {
      if (f < p) a = X[(long long)f * n + i];
      else if (f == p) a = y[i];
    }
    out[idx] = a;
  }
}

// Data-only sums of the canonical logistic form (include/fmh_detmath.h, fmh_logit_g; oracle: logit_hs): hs[0] = sum_i w_i when
// the model has an intercept, hs[ic + j] = sum_i w_i x_ij, w_i = +1/2 (y_i != 0) or -1/2 -- every product exact, the sums over
// the 512 canonical lanes in index order and their tree -- and behind them, for the range check of the fast loops, the largest
// |x| of every column (a NaN stays).  One workgroup, once per launch: n (p + 1) additions.
__global__ __launch_bounds__(NT) void logit_hs_kernel(const double* X, const double* y, long long n, int p, int ic, double* hs) {
  __shared__ double s_w[NW], s_m[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int col = -ic; col < p; col++) {
    double acc = 0.0, mx = 0.0;
    for (long long i = tid; i < n; i += NT) {
      const double w = (y[i] != 0.0) ? 0.5 : -0.5;
      if (col < 0) { acc = acc + w; }
      else {
        const double x = X[(long long)col * n + i], ax = __builtin_fabs(x);
        acc = acc + w * x;
        mx = (ax > mx || ax != ax) ? ax : mx;
      }
    }
    const double v = wave_xor_sum(acc);
    for (int o = 32; o >= 1; o >>= 1) { const double t = __shfl_xor(mx, o, 64); mx = (t > mx || t != t) ? t : mx; }
    if (lane == 0) { s_w[wave] = v; s_m[wave] = mx; }
    __syncthreads();
    if (tid == 0) {
      hs[ic + col] = ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) + ((s_w[4] + s_w[5]) + (s_w[6] + s_w[7]));
      if (col >= 0) {
        double m = s_m[0];
        for (int q = 1; q < NW; q++) m = (s_m[q] > m || s_m[q] != s_m[q]) ? s_m[q] : m;
        hs[ic + p + col] = m;
      }
    }
    __syncthreads();
  }
}
// per-workgroup slices for the observation-sharded logistic evaluation (mh_common.hpp, logit_shard): workgroup b owns the
// canonical lanes 2 b, 2 b + 1; xs[((b nslots + slot) 2 + q) p + j] = x_ij of observation i = 512 slot + 2 b + q (0 beyond n)
__global__ void logit_build_slices(const double* X, long long n, int p, int nslots, double* xs) {
  const int b = blockIdx.x;
  for (int idx = threadIdx.x; idx < nslots * 2 * p; idx += blockDim.x) {
    const int o = idx / p, j = idx - o * p;
    const long long i = (long long)NT * (o >> 1) + 2 * b + (o & 1);
    xs[(long long)b * nslots * 2 * p + idx] = (i < n) ? X[(long long)j * n + i] : 0.0;
  }
}

// the slices of the long-data form (mh_common.hpp, shard_long): per workgroup [p + 1][2 nslots], columns then y, observation
// o = 2 slot + q <-> i = 512 slot + 2 b + q (0 beyond n): thread t of the workgroup reads element o = t, t + 512, .. of every column
__global__ void long_build_slices(const double* X, const double* y, long long n, int p, int nslots, double* xs) {
  const int b = blockIdx.x, nobs = 2 * nslots;
  for (long long idx = threadIdx.x; idx < (long long)(p + 1) * nobs; idx += blockDim.x) {
    const int j = (int)(idx / nobs), o = (int)(idx - (long long)j * nobs);
    const long long i = (long long)NT * (o >> 1) + 2 * b + (o & 1);
    xs[(long long)b * (p + 1) * nobs + idx] = (i < n) ? (j < p ? X[(long long)j * n + i] : y[i]) : 0.0;
  }
}

// compact per-workgroup slices of X and y for the observation-sharded evaluation (mh_common.hpp, eval_sharded):
// xs[(b p + j) SH_MAXO + o], ys[b SH_MAXO + o] with o = slot * LPW + q <-> observation b LPW + q + 512 slot (0 beyond n)
__global__ void shard_build_slices(const double* X, const double* y, long long n, int p, int lpw, int nslots,
                                   double* xs, double* ys) {
  const int b = blockIdx.x;
  for (int idx = threadIdx.x; idx < (p + 1) * SH_MAXO; idx += blockDim.x) {
    const int j = idx / SH_MAXO, o = idx - j * SH_MAXO;
    const int sl = o / lpw, q = o - sl * lpw;
    const long long i = (long long)b * lpw + q + (long long)NT * sl;
    const bool valid = sl < nslots && i < n;
    if (j < p) xs[((long long)b * p + j) * SH_MAXO + o] = valid ? X[(long long)j * n + i] : 0.0;
    else ys[(long long)b * SH_MAXO + o] = valid ? y[i] : 0.0;
  }
}

// the same slices in fp64-MFMA operand layout (mh_common.hpp, shard_columns_mfma): per workgroup a block of
// shm_hdr(nmt) + nmt KB 64 doubles = validity bits | y in D layout | A tiles [mt][kb][lane]
// t10: the third M-tile in the layout of the two 4x4x4 MFMAs that compute its 8 live rows, per K-block 32 doubles [kk][i][r]
// = row 4 r + i of the tile (value t = 8 + r of lane group i), column 4 kb + kk; the other 32 doubles of the K-block stay 0
__global__ void shard_build_mfma(const double* X, const double* y, long long n, int p, int lpw, int nslots, int nmt, int t10,
                                 double* out, int blk_doubles) {
  const int b = blockIdx.x, KB = (p + 3) >> 2, H = 4 / lpw, spg = (nslots + H - 1) / H, HDR = shm_hdr(nmt);
  double* o = out + (long long)b * blk_doubles;
  auto obs_of = [&](int g, int t) -> long long {   // observation at D position (lane group g, value t), -1: none
    const int q = g / H, h = g % H;
    if (t >= spg) return -1;
    const int slot = spg * h + t;
    if (slot >= nslots) return -1;
    const long long i = (long long)b * lpw + q + (long long)NT * slot;
    return i < n ? i : -1;
  };
  for (int idx = threadIdx.x; idx < blk_doubles; idx += blockDim.x) {
    if (idx < 32) {
      unsigned w[2];
      for (int e = 0; e < 2; e++) {
        const int g = (2 * idx + e) >> 4;
        unsigned m = 0;
        for (int t = 0; t < SHM_T; t++) if (obs_of(g, t) >= 0) m |= 1u << t;
        w[e] = m;
      }
      ((unsigned*)o)[2 * idx] = w[0];
      ((unsigned*)o)[2 * idx + 1] = w[1];
    } else if (idx < HDR) {
      const int t = (idx - 32) >> 6, lane = (idx - 32) & 63;
      const long long i = obs_of(lane >> 4, t);
      o[idx] = i >= 0 ? y[i] : 0.0;
    } else {
      const int e = idx - HDR, lane = e & 63, kb = (e >> 6) % KB, mt = (e >> 6) / KB;
      if (t10 && mt == 2) {
        const int kk4 = lane >> 3, i4 = (lane >> 1) & 3, r4 = lane & 1, col4 = 4 * kb + kk4;
        const long long i = lane < 32 ? obs_of(i4, 8 + r4) : -1;
        o[idx] = (i >= 0 && col4 < p) ? X[(long long)col4 * n + i] : 0.0;
        continue;
      }
      const int row = lane & 15, kk = lane >> 4, col = 4 * kb + kk;
      const long long i = obs_of(row & 3, 4 * mt + (row >> 2));   // D register r of lane group g is row 4 r + g of the tile
      o[idx] = (i >= 0 && col < p && mt < nmt) ? X[(long long)col * n + i] : 0.0;
    }
  }
}

// accept counts of a continuation window (step windows, launch_sweep) added to the call's
__global__ void add_counts_kernel(long long* total, const long long* part, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) total[i] += part[i];
}

size_t sweep_lds_bytes(int k, int kf, int kind, int CW, int tb, int kz, bool resident) {
  size_t d = 5 * (size_t)k + (k / 2 + 1) + (size_t)NW * CW + 1 + (size_t)CW * tb * (kz + 1) + (size_t)CW * k +
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
static thread_local const char* g_kernel = "";
const char* fmcmc_last_kernel(void) { return g_kernel; }

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
  if (kn->k > FMCMC_MAX_K_WAVE) {   // one workgroup per chain (mh_sweep_bigk): what it implements
    const bool simple_joint = (kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE || kn->kind == FMCMC_KERNEL_UNIF || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE) &&
                              kn->scheme == FMCMC_SCHEME_JOINT;
    const bool adapt_plain = kn->kind == FMCMC_KERNEL_ADAPT && kn->bw == 0 && kn->freq <= 1;
    if (!(simple_joint || adapt_plain || kn->kind == FMCMC_KERNEL_RAM)) {
      set_err("k = %d > %d parameters: supported are kernel_normal(_reflective) / kernel_unif(_reflective) with scheme = 'joint', "
              "kernel_adapt(bw = 0, freq = 1) and kernel_ram", kn->k, FMCMC_MAX_K_WAVE);
      return FMCMC_ERR_UNSUPPORTED;
    }
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
  if (kn->kind == FMCMC_KERNEL_RAM) {
    if (kn->ram_qfun < FMCMC_RAM_QFUN_T_K || kn->ram_qfun > FMCMC_RAM_QFUN_T_DF) {
      set_err("kernel_ram: unknown -qfun- family %d (0 = rt(k, k), 1 = rnorm(k), 2 = rt(k, df)).", kn->ram_qfun);
      return FMCMC_ERR_ARG;
    }
    if (kn->ram_qfun == FMCMC_RAM_QFUN_T_DF && !(kn->ram_df > 0.0 && kn->ram_df <= DBL_MAX)) {
      set_err("kernel_ram: -qfun- = rt(k, df) needs a finite df > 0.");
      return FMCMC_ERR_ARG;
    }
    if (!(kn->ram_eta_exp >= 0.0 && kn->ram_eta_exp <= DBL_MAX)) {
      set_err("kernel_ram: the exponent of -eta- must be finite and positive (0 selects the default 2/3).");
      return FMCMC_ERR_ARG;
    }
  }
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

// The same sweep for the chains [off, off + cnt) of a call: every per-chain array advanced, RNG ids continued.
static SweepArgs chain_window(const SweepArgs& A, long long off, long long cnt, int kf) {
  SweepArgs W = A;
  const long long k = A.k, S = A.ldS, ns = A.nsteps, words = (A.nsteps + 31) >> 5;
  W.nchains = cnt; W.chain_base = A.chain_base + off;
#define ADV(f, stride) if (W.f) W.f += off * (stride)
  ADV(scheme_cols, ns); ADV(mirror_mu, k); ADV(mirror_scale, k); ADV(obs_arate, k);
  ADV(hist, (long long)A.hist_rows * kf);
  ADV(fed_logu, ns); ADV(fed_z, ns * A.kz);
  ADV(win_sum, kf); ADV(theta0, k); ADV(f0, 1); ADV(abs_iter, 1); ADV(Sigma, (long long)kf * kf); ADV(mean_prev, kf); ADV(have_mean, 1); ADV(nerrors, 1);
  ADV(samples, k * S); ADV(logpost, S); ADV(draws, k * S); ADV(accept_count, 1); ADV(accept_bits, words);
  ADV(status, 1); ADV(status_step, 1); ADV(status_theta, k);
#undef ADV
  return W;
}

// Observation-sharded evaluation (mh_common.hpp, eval_sharded): `nb` workgroups per launch must split the 512 canonical
// lanes evenly (128 or 256 of them), be co-resident (cooperative launch) and hold their slice in SH_MAXO registers.
// Returns the canonical lanes per workgroup (2 or 4), or 0 when the shape is not eligible or the cost model prefers the
// chain-sharded kernel.  Cost model (us per step, fitted at k = 50): chain-sharded ~4 + X bytes / 65 GB/s (the per-CU L2
// rate); sharded ~14 of hand-overs and fixed work + 0.0085 per column and walked observation slot (+ ~6 of barrier
// imbalance under kernel_ram): n = 2500 loses (23.9 vs 19.1), n = 5000 wins (24.3 vs 30.1), C4 wins 2x.
// Knob shard=1 (FMCMC_AMD_DEBUG) forces the sharded kernel for every eligible shape (tests), shard=0 disables it.
// ---- diagnosis knobs: ONE environment variable, read once per call --------------------------------------------------------
//   FMCMC_AMD_DEBUG="key=value,key=value"   (unset = product behaviour; nothing else in the environment is looked at)
//   streamed=1   general streamed kernel for everything          cw=1|2|4|8  chains per workgroup of the streamed kernels
//   pipe=0       no materialised-stream kernels (mfma / spec)      lat=0|1|2|3 latency form of mh_sweep_spec: off / chains per workgroup
//   mfma=0       VALU evaluation instead of the fp64-MFMA kernels
//   shard=0|1    wide models: never / always (when eligible) observation-sharded; unset: cost model
//   shard_mfma=0 VALU form of the sharded slice product           wide2=0|1   never / always (when eligible) the dataflow form
//   groups=4     four chain groups in the dataflow form (default two)       tiles=0     even N-tile shares of its evaluator waves
//   window=N     step-window length of the stream-fed kernels (multiple of 32; default: ~256 MiB of stream per window)
//   t10=0        the sharded slice product's third M-tile as a 16x16x4 tile even where 8 of its rows are padding
//   shadow=0     logistic, observation-sharded: the normal / uniform kernels on the general kernel's form, not on mh_sweep_logit2
//   speclogit=0  logistic family: not on the wave-specialised kernel (mh_sweep_spec<.., LOGISTIC>)
//   specbnd=0    the bounded kernel_ram: not on the wave-specialised kernel (SpecSyncB)
//   specmirror=0 the mirror kernels: not on the wave-specialised kernel
//   tinymfma=0   the streamed MFMA forms (8 .. 15 covariates, mirror / adaptive kernels) only from 513 observations on
//   specwide=0   kernel_adapt / kernel_ram with 8 .. 14 covariates on small data: not on the wave-specialised kernel
//   specp0=0     models without a covariate (iid Normal): adaptive / mirror kernels not on the wave-specialised kernel
//   turn=<t>     logit_shard's issue-priority turn (timing only): thousandths of the younger wave's passes it starts from, + 10000: and
//                stays at, + 100000 x (lead in units of 256 cycles it is regulated towards); turn=0: no turn
//   mode=<bits>  timing ablations and stamps (SweepArgs.debug)
// The kernel a call ended up on is reported by fmcmc_last_kernel(); DESIGN.md section 5 has the shape -> kernel table.
struct Knobs {
  int streamed = -1, cw = -1, pipe = -1, lat = -1, mfma = -1, shard = -1, shard_mfma = -1, wide2 = -1, groups = -1, tiles = -1, t10 = -1, window = -1, mode = 0;
  int shadow = -1, turn = -1, speclogit = -1, specbnd = -1, specmirror = -1, specp0 = -1, tinymfma = -1, specwide = -1;
};
static Knobs read_knobs() {
  Knobs K;
  const char* e = getenv("FMCMC_AMD_DEBUG");
  if (!e) return K;
  struct { const char* name; int* dst; } tab[] = {{"streamed", &K.streamed}, {"cw", &K.cw}, {"pipe", &K.pipe}, {"lat", &K.lat},
      {"mfma", &K.mfma}, {"shard_mfma", &K.shard_mfma}, {"shard", &K.shard}, {"wide2", &K.wide2}, {"groups", &K.groups}, {"tiles", &K.tiles}, {"t10", &K.t10}, {"window", &K.window}, {"mode", &K.mode}, {"shadow", &K.shadow}, {"turn", &K.turn}, {"speclogit", &K.speclogit}, {"specbnd", &K.specbnd}, {"specmirror", &K.specmirror}, {"specp0", &K.specp0}, {"tinymfma", &K.tinymfma}, {"specwide", &K.specwide}};
  while (*e) {
    const char* eq = strchr(e, '=');
    const char* end = strchr(e, ',');
    if (!end) end = e + strlen(e);
    if (eq && eq < end)
      for (auto& t : tab)
        if ((size_t)(eq - e) == strlen(t.name) && !strncmp(e, t.name, (size_t)(eq - e))) *t.dst = atoi(eq + 1);
    e = (*end == ',') ? end + 1 : end;
  }
  return K;
}
static bool shard_mfma_enabled(const Knobs& K) { return K.shard_mfma != 0; }
static int wide_sharded_lanes(const Knobs& K, const fmcmc_model* m, const fmcmc_kernel* kn, const fmcmc_run* run, int ram_bounded, int ncu, long long nb,
                              int cw_now = 2 /* chains per workgroup the call would run with on the chain-sharded / general kernel */) {
  if (K.shard == 0) return 0;
  if (m->family != FMCMC_FAM_GAUSSIAN_LINREG || m->p < 16) return 0;
  if (kn->kind != FMCMC_KERNEL_RAM && kn->kind != FMCMC_KERNEL_NORMAL && kn->kind != FMCMC_KERNEL_NORMAL_REFLECTIVE) return 0;
  const int nslots = (int)((m->n + NT - 1) / NT);
  const int lpw = (nb == 128 || nb == 256) ? (int)(NT / nb) : 0;
  const long long per_launch = nb * 2;   // (upper bound of the chains of one launch: at most two per workgroup)
  // A slice of more than 49 columns (15.5 KB) no longer stays in the scalar cache: 2.1x per walked slot, still ahead for the
  // normal kernels (k = 64, n = 10k: 57 us per step against 78); kernel_ram stays chain-sharded there, its owner phase
  // dominates at that width and runs slower in the sharded instantiation (121 against 108).
  const bool cached = shard_mfma_enabled(K) || (size_t)m->p * SH_MAXO * sizeof(double) <= 15872;   // (the MFMA form keeps the slice in LDS)
  // (a slice holds up to SH_MAXO = 40 observations in the scalar / register form, up to 4 SHM_T = 96 -- six M-tiles -- in LDS for the
  //  matrix-core form: n <= 24,576 at 256 workgroups)
  const bool mf_ok = shard_mfma_enabled(K) && m->p <= 4 * SHM_KBMAX;
  const bool ok = lpw > 0 && !(kn->kind == FMCMC_KERNEL_RAM && (ram_bounded || !cached)) && lpw * nslots <= (mf_ok ? 4 * SHM_T : SH_MAXO) && nb <= ncu &&
                  (long long)m->p * SH_MAXO * nb < (1ll << 28) && (long long)(m->p + 1) * (per_launch + SH_PAD) < (1ll << 31) &&
                  run->nsteps < 30000000;   /* barrier epochs (2 per step) x workgroups per group stay below 2^32 */
  if (!ok) return 0;
  if (K.shard != 1) {
    // us per step, refitted to tools/dispatch_audit.py (profiles/r04_dispatch_audit.md: p = 16 .. 60, n = 1e3 .. 1e4, 64 .. 2048
    // chains): the chain-sharded kernel streams the data set per workgroup and pays kernel_ram's owner phase (~0.15 us per
    // parameter) in the open; the sharded forms cost ~9 us of hand-overs plus a slice product that grows with the chains of a
    // launch -- on the matrix cores p (0.08 + 0.00475 slice observations) per 512 chains -- and hide the RAM owners in the
    // dataflow form (more than 256 chains), pay ~0.17 us per parameter in the sequential one
    const bool ram = kn->kind == FMCMC_KERNEL_RAM;
    // (with four / eight chains per workgroup -- more than 512 / 1024 chains -- the data stream is shared by more chains but a
    //  step takes 1.3x / 2.4x as long (and the owners of a workgroup queue), and the workgroups run in rounds; the sharded sweep runs as consecutive launches)
    const double rounds = (double)((run->nchains + (long long)cw_now * ncu - 1) / ((long long)cw_now * ncu));
    const double launches = (double)((run->nchains + per_launch - 1) / per_launch);
    const double est_chain = (4.0 + (double)m->n * (double)m->p * 8.0 / 65000.0 * (cw_now >= 8 ? 2.4 : (cw_now == 4 ? 1.3 : 1.0)) +
                              (ram ? (cw_now <= 2 ? 0.12 : 0.075 * (double)cw_now) * (double)kn->k : 0.0)) * rounds;
    const double frac = (double)(run->nchains < per_launch ? run->nchains : per_launch) / 512.0;
    double est_shard;
    if (shard_mfma_enabled(K) && m->p <= 4 * SHM_KBMAX) {
      // (more than three M-tiles: the run-time K-block loop, +5 us; kernel_ram's owners are hidden by the dataflow form only -- more
      //  than 256 chains, at most three M-tiles --, else ~0.17 us per parameter for few chains, ~0.3 in full launches)
      const bool tall = lpw * nslots > SH_MAXO;
      // (round 5: the dataflow form for 256 chains and fewer too -- two chains per workgroup, half of the workgroups without chains:
      //  C4's shape at 256 / 128 / 64 chains 15.0 / 14.7 / 12.7 us per step against 20.6 / 18.0 / 17.1 on the sequential form)
      const bool hidden = ram && !tall && !kn->constr && K.wide2 != 0 && (run->nchains > 256 || cw_now == 2);
      // (refitted once more after the compile-time K-block counts of every width: ~10 of hand-overs, 0.4 + p (0.083 + 0.004 slice observations) per 512
      //  chains at up to three M-tiles; the dataflow form's kernel_ram runs ~2 us UNDER the normal kernels' sequential form)
      const bool tall_rt = tall && (m->p + 3) / 4 > 12;     // (tall slices beyond 12 K-blocks keep the run-time loop: ~5 us more)
      est_shard = 10.2 + (tall_rt ? 5.0 : 0.0) + frac * ((tall ? 0.6 : 0.4) + (double)m->p * ((tall ? 0.08 : 0.083) + (tall ? 0.00475 : 0.004) * (double)(lpw * nslots))) +
                  ((ram && !hidden) ? (run->nchains <= 256 ? 0.17 : 0.3) * (double)kn->k : 0.0);
      if (hidden) {   // (what the dataflow form hides is at most a quarter of its slice product)
        const double prod = frac * (0.4 + (double)m->p * (0.083 + 0.004 * (double)(lpw * nslots)));
        est_shard -= (prod * 0.25 < 2.0 * frac) ? prod * 0.25 : 2.0 * frac;
      }
    }
    else {
      const double walked = (cached ? 1.0 : 2.1) * ((lpw * nslots <= SH_MAXO / 2) ? SH_MAXO / 2 : SH_MAXO);
      est_shard = 14.0 + 0.0085 * (double)m->p * walked + (ram ? 6.0 : 0.0);
    }
    if (!(est_shard * launches < 0.95 * est_chain)) return 0;
  }
  return lpw;
}
static bool wide_sharded_pays(const Knobs& K, const fmcmc_model* m, const fmcmc_kernel* kn, const fmcmc_run* run, int ram_bounded, int ncu, long long nb, int cw_now) {
  return wide_sharded_lanes(K, m, kn, run, ram_bounded, ncu, nb, cw_now) > 0;
}

// stream-ordered scratch that is released on EVERY way out of launch_sweep
struct AsyncScratch {
  void* p = nullptr;
  hipStream_t s = nullptr;
  ~AsyncScratch() { if (p) (void)hipFreeAsync(p, s); }
};

// launch of a kernel handle (mh_kernels.hpp): every sweep kernel takes the launch's SweepArgs by value
static hipError_t launch_k(const void* kfn, long long grid, int block, size_t lds, hipStream_t stream, const SweepArgs& A) {
  if (!kfn) return hipErrorInvalidDeviceFunction;
  if (lds > 48 * 1024) {
    const hipError_t ea = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
  }
  void* kargs[] = {(void*)&A};
  return hipLaunchKernel(kfn, dim3((unsigned)grid), dim3((unsigned)block), kargs, lds, stream);
}

// kernel->fixed etc. are DEVICE pointers here; kf and bounds info come via `kf`/`ram_bounded`.
static int launch_sweep(const fmcmc_model* m_in, const fmcmc_kernel* kn_in, const fmcmc_run* run,
                        fmcmc_state* st, fmcmc_out* out, int kf, int ram_bounded, hipStream_t stream) {
  SweepArgs A;
  memset(&A, 0, sizeof(A));
  // iid Normal(mu, sigma) IS the Gaussian linear model with an intercept and no covariate -- the same canonical arithmetic in
  // every kernel and in the oracle (fmcmc_oracle.c: pp = 0, icc = 1) -- so it takes that model's fast paths instead of the
  // all-family kernel (tools/option_audit.py)
  fmcmc_model m_norm = *m_in;
  if (m_norm.family == FMCMC_FAM_IID_NORMAL) { m_norm.family = FMCMC_FAM_GAUSSIAN_LINREG; m_norm.p = 0; m_norm.intercept = 1; }
  const fmcmc_model* m = &m_norm;
  AsyncScratch hist_guard, ws_guard, shw_guard, wc_guard;
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
    hist_guard.p = A.hist; hist_guard.s = stream;
  }
  A.freq = kn->freq < 1 ? 1 : kn->freq; A.scheme_seq = kn->scheme_seq; A.scheme_len = kn->scheme_len;
  A.constr = (kn->kind == FMCMC_KERNEL_RAM) ? kn->constr : nullptr; A.scheme_cols = st->scheme_cols;
  A.family = m->family; A.p = m->p; A.intercept = m->intercept ? 1 : 0; A.guard = m->guard ? 1 : 0;
  A.n = m->n; A.X = m->X; A.y = m->y; A.prior_div = m->prior_div;
  A.kind = kn->kind; A.k = kn->k; A.scheme = kn->scheme; A.warmup = kn->warmup;
  A.until = kn->until; A.eps = kn->eps; A.arate = kn->arate;
  // kernel_ram's qfun / eta families (R/kernel_ram.R:67-68): df of the t variates (0 = normal) and the exponent of eta
  A.ram_df = (kn->ram_qfun == FMCMC_RAM_QFUN_NORMAL) ? 0.0 : (kn->ram_qfun == FMCMC_RAM_QFUN_T_DF ? kn->ram_df : (double)kf);
  A.ram_neg_exp = (kn->ram_eta_exp != 0.0) ? -kn->ram_eta_exp : (-2.0 / 3.0);
  A.mu = kn->mu; A.scale = kn->scale; A.lb = kn->lb; A.ub = kn->ub; A.fixed = kn->fixed;
  A.nchains = run->nchains; A.nsteps = run->nsteps; A.burnin = run->burnin; A.thin = run->thin;
  A.S = fmcmc_kept_rows(run->nsteps, run->burnin, run->thin);
  A.ldS = out->ld_rows > 0 ? out->ld_rows : A.S;
  if (A.ldS < A.S) { set_err("fmcmc_out.ld_rows (%lld) is smaller than the %lld kept rows of this call", (long long)out->ld_rows, (long long)A.S); return FMCMC_ERR_ARG; }
  A.chain_base = run->chain_base; A.step_base = run->step_base; A.seed = run->seed;
  A.rng_mode = run->rng_mode; A.fresh = st->fresh; A.ram_bounded = ram_bounded;
  A.kz = variates_per_step(kn, kf);
  A.fed_logu = run->fed_logu; A.fed_z = run->fed_z;
  const Knobs K = read_knobs();
  A.debug = K.mode;   // timing ablations only
  A.theta0 = st->theta0; A.f0 = st->f0; A.abs_iter = (long long*)st->abs_iter; A.Sigma = st->Sigma;
  A.mean_prev = st->mean_prev; A.have_mean = st->have_mean; A.nerrors = st->nerrors;
  A.samples = out->samples; A.logpost = out->logpost; A.draws = out->draws;
  A.accept_count = (long long*)out->accept_count; A.accept_bits = out->accept_bits;
  A.status = out->status; A.status_step = (long long*)out->status_step; A.status_theta = out->status_theta;

  // logistic family: the data-only sums of the linear part and the columns' largest |x| (logit_hs_kernel), once per launch
  AsyncScratch hs_guard;
  if (m->family == FMCMC_FAM_LOGISTIC) {
    double* hs = nullptr;
    hipError_t eh = hipMallocAsync((void**)&hs, sizeof(double) * (size_t)(2 * MAXK + 2), stream);
    if (eh != hipSuccess) { set_err("hipMallocAsync(logistic sums) failed: %s", hipGetErrorString(eh)); return FMCMC_ERR_DEVICE; }
    hs_guard.p = hs; hs_guard.s = stream;
    hipLaunchKernelGGL(logit_hs_kernel, dim3(1), dim3(NT), 0, stream, m->X, m->y, (long long)m->n, m->p, m->intercept ? 1 : 0, hs);
    A.lg_hs = hs;
  }
  // ---- launch geometry
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  {   // the code object is gfx950 only, and the hand-overs of the wide kernels rest on ITS cache behaviour (mh_common.hpp)
    static int arch_ok = -1;
    if (arch_ok < 0) {
      hipDeviceProp_t prop;
      arch_ok = (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
    }
    if (!arch_ok) { set_err("this library is built for gfx950 (MI355X); the current device is another architecture"); return FMCMC_ERR_DEVICE; }
  }
  // more parameters than a wavefront has lanes: one workgroup per chain (mh_bigk.hpp); fmcmc_validate has refused what it lacks
  if (kn->k > FMCMC_MAX_K_WAVE) {
    const size_t blds = sizeof(double) * bigk_lds_doubles(kn->k, kf, kn->kind);
    if (blds > 160 * 1024) { set_err("LDS budget exceeded (k=%d)", kn->k); return FMCMC_ERR_UNSUPPORTED; }
    g_kernel = "big-k";
    hipError_t eb = launch_k(fmh::k_bigk(), run->nchains, NT, blds, stream, A);
    if (eb == hipSuccess) eb = hipGetLastError();
    if (eb != hipSuccess) { set_err("HIP launch failed: %s", hipGetErrorString(eb)); return FMCMC_ERR_DEVICE; }
    return FMCMC_OK;
  }
  // register-resident variant: Gaussian linreg whose data fits the VGPR budget of 512 threads
  int res_p = -1, res_opt = 0;
  const bool force = K.streamed == 1;
  if (!force && m->family == FMCMC_FAM_GAUSSIAN_LINREG && !mirror) {
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
  bool wide_switched = false;   // two chains per workgroup BECAUSE the observation-sharded sweep pays: the decision stands below
  if (resident) {
    cw = 4;
  } else {
    while (cw < NW && (long long)cw * ncu < run->nchains) cw <<= 1;
    if (K.cw == 1 || K.cw == 2 || K.cw == 4 || K.cw == 8) cw = K.cw;   // diagnosis: chains per workgroup of the streamed kernel
    // wide linear models with more than two chains per CU: two chains per workgroup, so that the sweep can run as
    // consecutive observation-sharded launches of 2 x CUs chains each (below) when that pays off
    // (measured at k = 50, n = 10k: 1024 chains 63.8 us per step instead of 71.6 with four chains per workgroup; at 2048
    //  chains the general kernel with eight chains per workgroup is level, 123 vs 128, and keeps the sweep)
    else if (cw >= 4 && wide_sharded_pays(K, m, kn, run, ram_bounded, ncu, (long long)ncu, cw)) { cw = 2; wide_switched = true; }
    // kernel_ram on a wide model with ONE chain per CU or fewer: two per workgroup all the same, so that the sweep is eligible for the
    // dataflow form (mh_wide2.hpp: two chain groups half a step out of phase; workgroups without chains evaluate like the others)
    else if (cw == 1 && ncu == 256 && K.wide2 != 0 && m->family == FMCMC_FAM_GAUSSIAN_LINREG && m->p >= 16 && m->p <= 4 * SHM_KBMAX &&
             kn->kind == FMCMC_KERNEL_RAM && !ram_bounded && !kn->constr && shard_mfma_enabled(K) &&
             2 * ((m->n + NT - 1) / NT) <= SH_MAXO && run->nchains >= 2 &&
             !((run->nchains + 1) / 2 == 128 && 4 * ((m->n + NT - 1) / NT) <= SH_MAXO) &&      /* (exactly 128 workgroups of four lanes: the sequential form's own shape) */
             (m->n + NT - 1) / NT >= 6 &&      /* (short data: the chain-sharded sweep is ahead -- p = 30, n = 1000, 64 chains: 9.7 against 11.6 us) */
             [&]() {                           /* (the dataflow form's LDS: two owners' factor and partial sums + the slice block -- k = 62 does not fit) */
               const int nsl = (int)((m->n + NT - 1) / NT), spg = (nsl + 1) / 2, nmt_ = (spg + 3) / 4;
               const int mblk_ = shm_hdr(nmt_) + nmt_ * ((m->p + 3) / 4) * 64;
               return nmt_ >= 1 && nmt_ <= 3 && sizeof(double) * wide2_lds_doubles(kn->k, kf, kn->kind, A.kz, mblk_) <= 160 * 1024;
             }() &&
             wide_sharded_pays(K, m, kn, run, ram_bounded, ncu, (long long)ncu, 2)) { cw = 2; wide_switched = true; }
    // the logistic-only instantiations (table in LDS; observation-sharded form) exist for up to four chains per workgroup: more
    // than 1024 chains run as more workgroups / consecutive sharded launches there, not on the all-family kernel with eight
    // chains per workgroup (tools/dispatch_audit.py: 4096 chains, n = 1e5, p = 5 took 1244 us per step, 4.7x four launches)
    else if (cw > 4 && m->family == FMCMC_FAM_LOGISTIC && kn->kind >= FMCMC_KERNEL_NORMAL && kn->kind <= FMCMC_KERNEL_RAM) cw = 4;
  }
  int tb = 32;
  while (tb > 1 && sweep_lds_bytes(kn->k, kf, kn->kind, cw, tb, A.kz, resident) > 60 * 1024) tb >>= 1;
  while (cw > 1 && !resident && sweep_lds_bytes(kn->k, kf, kn->kind, cw, tb, A.kz, resident) > 150 * 1024) cw >>= 1;
  A.tb = tb;
  size_t lds = sweep_lds_bytes(kn->k, kf, kn->kind, cw, tb, A.kz, resident);
  if (lds > 160 * 1024) { set_err("LDS budget exceeded (k=%d)", kn->k); return FMCMC_ERR_UNSUPPORTED; }
  const long long nblk = (run->nchains + cw - 1) / cw;
  hipError_t e = hipSuccess;
  // software-pipelined fast path: normal kernels, joint scheme, k <= 16, linreg data in registers
  const bool nopipe = K.pipe == 0;
  int pipe_opt = 0, mfma_ng = 0, mfma_ad = 0, mfma_ext = 0, spec_cw = 4;
  // kernel_adapt(freq = 2 .. 8, bw = 0) on the register owner of mh_sweep_spec (round 5: the last `freq` rows of a chain in an LDS ring;
  // tools/option_audit.py had it on the general kernel at 14.7 us per step where freq = 1 takes 3.3): no fixed parameter, k <= 8, and a
  // call that is ONE step window (the ring does not travel between windows)
  bool adapt_ring = false;
  if (kn->kind == FMCMC_KERNEL_ADAPT && kn->bw == 0 && kn->freq >= 2 && kn->freq <= SPEC_FREQMAX && kf == kn->k && A.kz == kn->k && kn->k <= SPEC_KA) {
    const long long per_step = (long long)run->nchains * (A.kz + 1) * 8;
    long long win = ((256ll << 20) / (per_step > 0 ? per_step : 1)) & ~31ll;
    if (win < 32) win = 32;
    if (K.window >= 32) win = (long long)K.window & ~31ll;
    adapt_ring = run->nsteps <= win + 1 || run->rng_mode != FMCMC_RNG_PHILOX;
  }
  bool lat_normal = false;   // the normal / uniform kernels in the latency form (mh_sweep_lat)   // mfma_ext: resident slots of the EXT form (0: everything resident)
  // single-parameter schemes of the normal / uniform kernels ("ordered", an explicit sequence, "random"): on mh_sweep_lat's candidate
  // wave (round 5: they ran on the general kernel, 2.9 us per step at the README's size where the joint scheme takes 0.63), one to FOUR
  // chains per workgroup; "random" draws its plan in the kernel and hands it back (a caller-fed plan stays general)
  const bool single_lat = (kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) && kn->scheme != FMCMC_SCHEME_JOINT &&
                          K.lat != 0 && (kn->scheme != FMCMC_SCHEME_RANDOM || run->rng_mode == FMCMC_RNG_PHILOX);
  AsyncScratch mfs_guard;
  if (!force && !nopipe && m->family == FMCMC_FAM_GAUSSIAN_LINREG &&
      (kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE ||
       ((kn->kind == FMCMC_KERNEL_ADAPT && (!adapt_hist || adapt_ring)) || (kn->kind == FMCMC_KERNEL_RAM && !kn->constr)) ||
       (mirror && kn->scheme == FMCMC_SCHEME_JOINT && kf == kn->k && K.mfma != 0)) &&
      (kn->scheme == FMCMC_SCHEME_JOINT || kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM || single_lat) && kn->k <= PIPE_KMAX &&
      // Sizes (round 3: rows and variates are addressed as 64-bit chain base + 32-bit offset, and a long call runs as step
      // windows with a bounded stream, so a call no longer leaves these kernels at 4 GiB of samples or stream).  What is
      // left: offsets inside one chain's blocks are 32 bits.
      (unsigned long long)run->nsteps * (unsigned long long)A.kz * 8ull < (1ull << 32) && run->nsteps < (1ll << 30) &&
      (unsigned long long)kn->k * (unsigned long long)A.ldS * 8ull < (1ull << 32) &&      /* 32-bit offsets inside ONE chain's block */
      /* (round 5: kernel_adapt / kernel_ram run in step windows too -- their step-dependent rules read the CALL's step, see below;
          what still materialises its whole stream, kept below 8 GiB: host-fed variates are the caller's, and the mirror kernels) */
      (!mirror || (unsigned long long)run->nchains * (unsigned long long)run->nsteps * (unsigned long long)(A.kz + 1) * 8ull < (8ull << 30))) {
    // the wave-specialised kernel (mh_sweep_spec): x of a compute lane in VGPRs, the slot count an (even) run-time choice among
    // its compute loops: any n <= 10240 at p <= 3, n <= 5120 at p = 4, 5, n <= 4096 at p = 6, 7 (OPTMAX P doubles per lane)
    {
      const long long nsl = (m->n + NT - 1) / NT, nsl2 = (nsl + 1) & ~1ll;
      // (round 5: 8 .. 14 covariates on up to 2048 observations -- four slots of P doubles per compute lane, the register owner at the
      //  compile-time width k <= 16; knob specwide=0: the streamed MFMA evaluation with the owners in LDS, as before)
      const int optmax = (m->p >= 0 && m->p <= 3) ? 20 : (m->p <= 5 ? 10 : (m->p <= 7 ? 8 : ((m->p <= 14 && K.specwide != 0) ? 4 : 0)));
      // (the bounded kernel_ram decides on f of the REFLECTED proposal: a second evaluation in the steps in which the reflection
      //  moved something -- the barrier-synchronised owners of mh_sweep_mfma_ad ask for it between barriers, this kernel's register
      //  owners through a second evaluation slot per step (round 5, SpecSyncB: k <= 8, no fixed parameter; knob specbnd=0: off))
      const bool bnd_ok = K.specbnd != 0 && kf == kn->k && (kn->k <= SPEC_KA || (kn->k == 9 && m->p == 7)) && A.kz == kn->k && !kn->constr;   // (k = 9: the compile-time owner of p = 7)
      // (round 5: no covariate at all -- the iid Normal family, intercept + sigma -- too: the compute lanes then hold no x)
      if ((m->p >= 1 || (m->p == 0 && m->intercept && K.specp0 != 0)) && nsl2 <= optmax && (kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM) && (!(kn->kind == FMCMC_KERNEL_RAM && ram_bounded) || bnd_ok)) pipe_opt = (int)nsl2;
      // (normal / uniform kernels run on the MFMA kernel; knob mfma=0 keeps them here for the two shapes they were tuned at)
      if (m->p == 3 && nsl == 20 && kn->kind < FMCMC_KERNEL_ADAPT && kn->scheme == FMCMC_SCHEME_JOINT) pipe_opt = 20;
      if (m->p == 1 && nsl == 2 && kn->kind < FMCMC_KERNEL_ADAPT && kn->scheme == FMCMC_SCHEME_JOINT) pipe_opt = 2;
    }
    // (round 5: the streamed forms from ONE observation on -- up to 512 the one resident slot is the last, nothing is streamed; models with
    //  8 .. 15 covariates on small data ran on the general kernel, 2.4 - 5 / 9 - 26 us per step.  Knob tinymfma=0: from 513 on, as before)
    const long long nt_min = (K.tinymfma != 0) ? 0 : (long long)NT;
    // fp64-MFMA evaluation: general in n and p up to what 80 operand registers per lane hold (normal / uniform kernels)
    if (K.mfma != 0 && kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && kn->scheme == FMCMC_SCHEME_JOINT) {
      if (m->p <= 3 && m->n <= (long long)NT * 20) mfma_ng = 1;
      else if (m->p <= 7 && m->n <= (long long)NT * 10) mfma_ng = 2;
      // beyond the operand registers: 16 (8) slots resident, the rest streamed from an operand-order copy every step (EXT)
      else if (m->p <= 3 && m->n < (1ll << 29)) { mfma_ng = 1; mfma_ext = 16; }
      else if (m->p <= 7 && m->n < (1ll << 29)) { mfma_ng = 2; mfma_ext = 8; }
      // 8 .. 15 covariates (k <= 16): three / four operand groups per observation slot, four / two slots resident (one for short
      // data), the rest streamed -- tools/dispatch_audit.py found these models on the general kernel at 0.10 of the fp64 peak where
      // p = 7 runs at 0.44
      else if (m->p <= 11 && m->n > nt_min && m->n < (1ll << 29)) { mfma_ng = 3; mfma_ext = (m->n > (long long)NT * 4) ? 4 : 1; }
      else if (m->p <= 15 && m->n > nt_min && m->n < (1ll << 29)) { mfma_ng = 4; mfma_ext = (m->n > (long long)NT * 2) ? 2 : 1; }
      // (the wave-specialised VALU kernel, which overlaps owners and evaluation, used to win at its small shape
      //  (p = 1, n ~ 1000); since the instruction diet of the owner phase the MFMA kernel is 1.2-1.35x ahead there too:
      //  tools/bench_small.py.  Knob mfma=0 still selects it.)
    }
    // kernel_adapt / kernel_ram beyond mh_sweep_spec's registers: the same streamed MFMA evaluation with the register-row
    // adaptive owners between barriers (mh_mfma_ad.hpp); no fixed parameter, k <= 8
    // (k = 9 -- seven covariates, intercept and sigma -- as a compile-time row count: tools/dispatch_audit.py found these calls on
    //  the general kernel, 7x the time of the normal kernels at the same shape)
    // (mfma_ad == 2: the owners with their matrices in LDS -- 8 .. 15 covariates, or a fixed parameter; not the bounded kernel_ram)
    if (K.mfma != 0 && !pipe_opt && !adapt_hist && (kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM) && m->p >= 0 && m->p <= 15 && m->n < (1ll << 29)) {   // (p = 0: iid Normal)
      // (round 5: the run-time-width register owner takes fixed parameters -- free ones first, the fixed ones as passengers)
      const bool reg_owner = m->p <= 7 && A.kz == kf && kf >= 1 && ((kf == kn->k && (kn->k <= SPEC_KA || kn->k == 9)) || (kf < kn->k && kn->k <= SPEC_KA));
      const int ng = (m->p <= 3) ? 1 : (m->p <= 7 ? 2 : (m->p <= 11 ? 3 : 4));
      const int nsr = (ng == 1) ? MfmaAdShape<1>::NSR : (ng == 2 ? MfmaAdShape<2>::NSR : MfmaAdShape<3>::NSR);
      if (m->n > (long long)NT * nsr && (reg_owner || !(kn->kind == FMCMC_KERNEL_RAM && ram_bounded))) {   // (its resident slots are all full)
        mfma_ad = reg_owner ? 1 : 2;
        mfma_ng = ng;
        mfma_ext = nsr;
      } else if (m->n > nt_min && ((reg_owner && ((kn->kind == FMCMC_KERNEL_RAM && ram_bounded) || m->p == 0)) || (!reg_owner && !(kn->kind == FMCMC_KERNEL_RAM && ram_bounded) && run->nchains <= 2048 /* (beyond: level with the general kernel at eight chains per workgroup) */))) {
        // short data (one slot resident, the rest streamed) for what the wave-specialised kernel does not take: the bounded
        // kernel_ram, 8 .. 15 covariates, no covariate at all (iid Normal)
        mfma_ad = reg_owner ? 1 : 2;
        mfma_ng = ng;
        mfma_ext = 1;
      }
    }
    if (kn->kind == FMCMC_KERNEL_RAM && ram_bounded && !mfma_ad && !pipe_opt) mfma_ng = 0;   // (general kernel)
    // the mirror kernels (joint scheme, no fixed parameter): their owner between the barriers of the same streamed MFMA evaluation
    if (mirror) {
      const int ng = (m->p <= 3) ? 1 : (m->p <= 7 ? 2 : (m->p <= 11 ? 3 : 4));
      const int nsr = (ng == 1) ? MfmaAdShape<1>::NSR : (ng == 2 ? MfmaAdShape<2>::NSR : MfmaAdShape<3>::NSR);
      pipe_opt = 0; mfma_ng = 0;
      // (round 5: within mh_sweep_spec's registers their owner runs there -- beside the evaluation instead of between barriers, and in
      //  the latency forms; up to 512 observations they ran on the general kernel.  Knob specmirror=0: off)
      const long long nsl2 = (((m->n + NT - 1) / NT) + 1) & ~1ll;
      if (K.specmirror != 0 && (m->p >= 1 || (m->p == 0 && m->intercept && K.specp0 != 0)) && (m->p <= 7 || (m->p <= 14 && K.specwide != 0)) && nsl2 <= fmh::k_spec_optmax(m->p, kn->kind) && fmh::k_spec(m->p, kn->kind)) pipe_opt = (int)nsl2;
      else
      if (m->p <= 15 && m->n > nt_min && m->n < (1ll << 29)) { mfma_ad = 3; mfma_ng = ng; mfma_ext = (m->n > (long long)NT * nsr) ? nsr : 1; }
    }
    // ---- the LATENCY form (round 5): fewer than four chains per compute unit.  The reference scales a FIXED number of chains
    // over its workers (R/mcmc.R:536-641), and a sharded call leaves every GPU nchains / G of them: with four chains per
    // workgroup a step of C2's shape costs the same 2 us at 64 chains and at 1024.  Here the wave-specialised kernel runs one,
    // two or three chains per workgroup -- all eight compute waves on the chain(s) there are (an evaluation of n = 10,000 is
    // 0.33 us of one CU's fp64 issue), no owner queued behind the evaluation of other chains -- for every shape its compute
    // lanes hold in registers: kernel_adapt / kernel_ram on mh_sweep_spec (its owners no longer queue behind the evaluation of
    // other chains), the normal / uniform kernels on mh_sweep_lat (mh_lat.hpp: chain state replicated in every wave, ONE barrier
    // per step).  Same canonical lanes and tree: the bits do not depend on the form.  Knob lat=0: off; lat=1|2|3: forced.
    // (8 .. 15 covariates on up to 2048 observations, round 5: mh_sweep_lat<KIND, P, 4> -- the joint scheme with ONE chain per compute unit
    //  (1.2 us per step on the streamed MFMA form at any chain count), the single-parameter schemes up to four (general kernel before))
    const long long nsl2w = (((m->n + NT - 1) / NT) + 1) & ~1ll, per_cuw = (run->nchains + ncu - 1) / ncu;
    const bool wide_lat = K.lat != 0 && K.specwide != 0 && !mirror && m->p >= 8 && m->p <= 15 && kn->k <= PIPE_KMAX && nsl2w <= 4 &&
                          kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && kn->kind >= FMCMC_KERNEL_NORMAL && kf >= 1 && fmh::k_lat(m->p, kn->kind) != nullptr;
    if (wide_lat && kn->scheme == FMCMC_SCHEME_JOINT && (per_cuw <= 1 || (K.lat >= 1 && K.lat <= 3))) {
      pipe_opt = (int)nsl2w; mfma_ng = 0; mfma_ext = 0; lat_normal = true;
      spec_cw = (K.lat >= 1 && K.lat <= 3) ? K.lat : 1;
    } else
    if (single_lat && !mirror) {
      const long long per_cu = (run->nchains + ncu - 1) / ncu, nsl2 = (((m->n + NT - 1) / NT) + 1) & ~1ll;
      if (per_cu <= 4 && m->p >= 0 && (m->p <= 7 ? nsl2 <= fmh::k_spec_optmax(m->p, kn->kind) : wide_lat) && fmh::k_lat(m->p, kn->kind)) {
        pipe_opt = (int)nsl2; mfma_ng = 0; lat_normal = true;
        spec_cw = (K.lat >= 1 && K.lat <= 3) ? K.lat : (int)per_cu;
      }
    } else
    if (K.lat != 0 && (!mirror || pipe_opt) && (pipe_opt || (mfma_ng && !mfma_ext && !mfma_ad))) {
      const long long per_cu = (run->nchains + ncu - 1) / ncu;
      // kernel_adapt / kernel_ram (mh_sweep_spec) gain up to 25 % with one chain per workgroup, 18 % with two, 6 % with three at
      // n = 10,000 and are level at small n -- their step is the owner's dependent chain --: one to three, always.  The normal
      // kernels by a cost model (us per step, fitted to tools/bench_lat_grid.sh and `tools/dispatch_audit.py --only=few`,
      // profiles/r05_dispatch_audit_few.md): mh_sweep_lat costs ~0.45 us of fold, barrier and decision plus, per chain of the
      // workgroup, its evaluation (n (p + 2) fp64 instructions at ~4.7 cycles over four SIMDs; shorter lanes of p >= 4 run
      // at a lower rate) or -- short data -- its coefficient broadcast and tree; the MFMA kernel's four chains cost ~0.8 us
      // + 0.06 us per operand group and observation slot.  n = 10,000, p = 3: 1.03 | 1.62 | 2.15 us with 1 | 2 | 3 chains
      // against 2.0; p = 1: three chains still win (1.54 against 2.07); p = 7, n = 1000: two lose (1.22 against 1.13).
      int lcw_auto = 4;
      if (per_cu <= 3) {
        if (kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE) {
          const double w = (double)m->n * (double)(m->p + 2), rate = (m->p <= 3) ? 9.2e-6 : 1.25e-5;
          const double per_chain = (0.10 + rate * w > 0.18 + 0.025 * (double)m->p) ? 0.10 + rate * w : 0.18 + 0.025 * (double)m->p;
          const double t_lat = 0.45 + (double)per_cu * per_chain;
          const double ns = (double)((m->n + NT - 1) / NT), ng = (m->p <= 3) ? 1.0 : 2.0;
          const double t_floor = 0.98 + 0.10 * (ng - 1.0);
          const double t_mfma = (0.80 + 0.06 * ng * ns > t_floor) ? 0.80 + 0.06 * ng * ns : t_floor;
          if (t_lat < t_mfma) lcw_auto = (int)per_cu;
        } else {
          lcw_auto = (int)per_cu;
        }
      }
      const int lcw = (K.lat >= 1 && K.lat <= 3) ? K.lat : lcw_auto;
      const long long nsl2 = (((m->n + NT - 1) / NT) + 1) & ~1ll;
      // (p = 0 -- the iid Normal family -- included: the compute lanes then hold no x)
      if (lcw < 4 && nsl2 <= fmh::k_spec_optmax(m->p, kn->kind)) {
        if (kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE) { pipe_opt = (int)nsl2; mfma_ng = 0; lat_normal = true; }
        if (pipe_opt && !mfma_ng) spec_cw = lcw;
      }
    }
  }
  // ---- the logistic family on the wave-specialised kernel (round 5; mh_spec.hpp, FAM = LOGISTIC): data in the compute lanes'
  // registers, g table in LDS, the register owners.  The workflow vignette's own model (mcmc::logit: 100 observations, k = 5) ran
  // on the general kernel at 2.6 / 5.9 us per step (kernel_normal / kernel_adapt).  Knob speclogit=0: off.
  bool spec_logit = false;
  // (a fixed parameter under the normal / uniform kernels: the latency form's candidate wave handles it, the owners of mh_sweep_spec do not)
  const bool lg_lat_fixed = kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && kn->kind >= FMCMC_KERNEL_NORMAL && kn->scheme == FMCMC_SCHEME_JOINT &&
                            kf != kn->k && K.lat != 0 && K.speclogit != 2;
  // (8 .. 15 covariates, k <= 16, up to 2048 observations, the normal / uniform kernels: the latency form only -- four slots of P doubles
  //  per lane; they ran on the general kernel, 3 - 4.5 us per step at n = 200)
  const bool lg_lat_wide = m->p >= 8 && m->p <= 15 && kn->k <= PIPE_KMAX && kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && kn->kind >= FMCMC_KERNEL_NORMAL &&
                           K.lat != 0 && K.speclogit != 2 && kf >= 1 && (kn->scheme == FMCMC_SCHEME_JOINT || single_lat);
  // (and under kernel_adapt / kernel_ram -- unbounded, stride 1, no fixed parameter --: mh_sweep_spec<P, 4, KIND, LOGISTIC> with the register
  //  owner at the compile-time width k <= 16; general kernel: 6 - 14 us per step at n = 200.  Knob specwide=0: off)
  const bool lg_spec_wide = m->p >= 8 && m->p <= 15 && kn->k <= PIPE_KMAX && K.specwide != 0 &&
                            ((kn->kind == FMCMC_KERNEL_ADAPT && !adapt_hist) || (kn->kind == FMCMC_KERNEL_RAM && !ram_bounded && !kn->constr));
  if (!force && !nopipe && K.speclogit != 0 && K.shard < 0 && m->family == FMCMC_FAM_LOGISTIC && !mirror && m->p >= 1 && (m->p <= 7 || lg_lat_wide || lg_spec_wide) &&
      kn->k == m->p + (m->intercept ? 1 : 0) && ((kf == kn->k && A.kz == kn->k) || single_lat || (lg_lat_fixed && kf >= 1)) &&
      (((kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE) && (kn->scheme == FMCMC_SCHEME_JOINT || single_lat)) ||
       (kn->kind == FMCMC_KERNEL_ADAPT && (!adapt_hist || adapt_ring)) || (kn->kind == FMCMC_KERNEL_RAM && !kn->constr && (!ram_bounded || K.specbnd != 0))) &&
      (unsigned long long)run->nsteps * (unsigned long long)A.kz * 8ull < (1ull << 32) && run->nsteps < (1ll << 28) &&
      (unsigned long long)kn->k * (unsigned long long)A.ldS * 8ull < (1ull << 32)) {
    const long long nsl2 = (((m->n + NT - 1) / NT) + 1) & ~1ll;
    const long long per_cu = (run->nchains + ncu - 1) / ncu;
    if ((kn->scheme != FMCMC_SCHEME_JOINT || lg_lat_fixed || lg_lat_wide) && kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE) {
      // single-parameter schemes: the latency form's candidate wave, one to four chains per workgroup (as for the linear model above)
      if (per_cu <= 4 && nsl2 <= (lg_lat_wide ? 4 : (lg_lat_fixed && fmh::k_spec_optmax(m->p, kn->kind) > 12 ? 12 : fmh::k_spec_optmax(m->p, kn->kind))) && fmh::k_lat_logit(m->p, kn->kind)) {   // (never beyond the kernel's own slot count: the randomised soak's case 1566)
        pipe_opt = (int)nsl2; spec_logit = true; lat_normal = true;
        spec_cw = (K.lat >= 1 && K.lat <= 3) ? K.lat : (int)per_cu;
      }
    } else
    if (nsl2 <= fmh::k_spec_optmax(m->p, kn->kind) && fmh::k_spec_logit(m->p, kn->kind)) {
      pipe_opt = (int)nsl2;
      spec_logit = true;
      spec_cw = (K.lat >= 1 && K.lat <= 3) ? K.lat : ((K.lat != 0 && per_cu <= 3) ? (int)per_cu : 4);
      // the normal / uniform kernels with fewer than four chains per CU: the latency form (mh_sweep_lat<.., LOGISTIC>: replicated decision)
      // (measured, tools/bench_small_logit.py and the pair of forms at 256 / 512 / 768 chains: the replicated decision wins up to ~3,000
      //  observations at any count -- 0.85 / 1.18 / 1.59 us against 1.25 / 1.34 / 1.78 at n = 1000 -- and up to ~6,000 with one chain
      //  per workgroup, 1.66 against 2.15 at n = 5000; beyond, the lookups' LDS time is the step and the owners' overlap pays:
      //  n = 10,000: 6.4 against 4.8 at 512 chains.  Knob speclogit=2: never.)
      if (spec_cw < 4 && K.speclogit != 2 && kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && fmh::k_lat_logit(m->p, kn->kind) &&
          nsl2 <= (spec_cw == 1 ? 12 : 6)) lat_normal = true;
    }
  }
  // (a launch with a slot count its instantiation does not hold would run no loop and hand back zeros -- round 5's soak found one such
  //  route --: whatever the rules above decided, a count beyond the kernel's own goes to the general kernels)
  if (pipe_opt && !mfma_ng) {
    const int own = (m->p <= 3) ? 20 : (m->p <= 5 ? 10 : (m->p <= 7 ? 8 : 4));
    if (pipe_opt > own || (pipe_opt & 1)) {
      if (A.debug) fprintf(stderr, "fmcmc_amd: slot count %d beyond the register kernels' %d at p = %d: general kernel\n", pipe_opt, own, m->p);
      pipe_opt = 0; lat_normal = false; spec_logit = false; spec_cw = 4;
    }
  }
  A.spec_opt = pipe_opt;
  A.spec_cw = spec_cw;
  A.nsteps_call = run->nsteps;
  // ---- the LONG-DATA form (mh_common.hpp, shard_long): few chains on long data.  Up to four chains are one workgroup of the
  // chain-sharded kernels, i.e. ONE compute unit walks the whole data set per step (n = 1e5, p = 3: 34 us per step, 255 CUs idle);
  // here all 256 workgroups evaluate their 1/256 of the observations for every chain and the canonical lane sums cross the chip as
  // in the other observation-sharded forms.  try_long(kernel, its LDS bytes without the term block, what the call costs otherwise)
  // launches it when the cost model -- or knob shard=1 -- says so; us per step, fitted on `tools/dispatch_audit.py --only=long`
  // (profiles/r04_dispatch_audit.md): ~8 us of hand-overs, the walk of a lane's slots (1.6e-5 us per observation; sums of the
  // logistic terms 1.0e-5) once per group of chains whose terms fit the LDS, and per chain its terms and its share of the exchange.
  bool launched_long = false;
  auto try_long = [&](const void* kfn, size_t lds_base, double est_now, bool logistic) -> int {
    if (force || K.shard == 0 || cw != 1 || ncu != 256 || run->nchains > 64 || m->n < 8 * NT || m->n >= (1ll << 31) || run->nsteps >= 30000000 ||
        (kn->kind == FMCMC_KERNEL_RAM && ram_bounded) || kn->kind < FMCMC_KERNEL_NORMAL || kn->kind > FMCMC_KERNEL_RAM) return FMCMC_OK;
    const int nslots = (int)((m->n + NT - 1) / NT), nobs = 2 * nslots;
    const long long room = ((long long)150 * 1024 - (long long)lds_base) / 8 - 2;
    const long long lrow = 2ll * shard_long_row(nslots) + SHL_BS;     // LDS doubles per chain of a group
    long long lcg = room / lrow;
    if (lcg > run->nchains) lcg = run->nchains;
    if (lcg < 1) return FMCMC_OK;
    const double pn = (double)m->n;
    const double groups = (double)((run->nchains + lcg - 1) / lcg);
    const double est_long = 8.3 + groups * (logistic ? 1.0e-5 : 1.6e-5) * pn + (logistic ? 2.0e-6 : 0.5e-6) * pn * (double)(m->p + 1) +
                            (double)run->nchains * (0.17 + 0.028 * (double)m->p + (logistic ? 3.0e-6 : 1.2e-6) * pn) +
                            (kn->kind >= FMCMC_KERNEL_ADAPT ? 3.5 : 0.0);
    if (!(K.shard == 1 || est_long < 0.9 * est_now)) return FMCMC_OK;
    const size_t llds = lds_base + sizeof(double) * (size_t)(lcg * lrow + 2);
    int coop = 0, perCU = 0;
    (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    hipError_t el = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds);
    if (el != hipSuccess || !coop || hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kfn, NT, llds) != hipSuccess || perCU < 1) return FMCMC_OK;
    double* shw = nullptr;
    const size_t nxs = (size_t)256 * (m->p + 1) * nobs, nth = ((size_t)kn->k * (run->nchains + SH_PAD) + 7) & ~(size_t)7,
                 npt = (size_t)(NT + SH_PAD) * run->nchains, nbar = 32 * 20 / 2;
    hipError_t ea = hipMallocAsync((void**)&shw, sizeof(double) * (nxs + nth + npt + nbar), stream);
    if (ea != hipSuccess) { set_err("hipMallocAsync(sharded evaluation) failed: %s", hipGetErrorString(ea)); return FMCMC_ERR_DEVICE; }
    shw_guard.p = shw; shw_guard.s = stream;
    double* thw = shw; double* ptw = thw + nth; unsigned* bar = (unsigned*)(ptw + npt); double* xs = ptw + npt + nbar;
    hipLaunchKernelGGL(long_build_slices, dim3(256), dim3(256), 0, stream, m->X, m->y, (long long)m->n, m->p, nslots, xs);
    (void)hipMemsetAsync(bar, 0, sizeof(double) * nbar, stream);
    SweepArgs W = A;
    W.shard = 2; W.sh_nslots = nslots; W.sh_xs = xs; W.sh_ys = nullptr; W.sh_th = thw; W.sh_part = ptw; W.sh_bar = bar;
    W.sh_long = 1; W.sh_lcg = (int)lcg; W.sh_mblk = (int)(lcg * lrow);
    void* kargs[] = {(void*)&W};
    hipError_t ec = hipLaunchCooperativeKernel(kfn, dim3(256), dim3(NT), kargs, (unsigned int)llds, stream);
    if (ec == hipSuccess) { launched_long = true; g_kernel = "long-sharded"; }
    else {                            // the runtime refused the cooperative launch: nothing ran, take the usual kernels
      (void)hipGetLastError();
      (void)hipFreeAsync(shw, stream);   // (the paths below put their own block into shw_guard)
      shw_guard.p = nullptr;
    }
    return FMCMC_OK;
  };
  // (wide linear models, p >= 16: where the matrix-core slices end -- 96 observations per workgroup, n = 24,576 -- the chain-sharded
  //  kernel is what is left: 4 + n p 8 / 65000 us per step)
  if (m->family == FMCMC_FAM_GAUSSIAN_LINREG && (m->p <= 15 || (m->p <= 62 && m->n > (long long)NT * 2 * SHM_T))) {
    const double pn = (double)m->n;
    const double now_rate = (m->p <= 3) ? (pn <= 2e5 ? 3.3e-4 : 5.1e-4) : (m->p <= 7 ? 4.9e-4 /* (round 5 audit: n = 2e4, p = 7, one chain: 9.75 us on the streamed MFMA kernel, the long-data form 10.6) */ : (m->p <= 11 ? 8.5e-4 : 1.17e-3));
    const double est_now = (m->p >= 16) ? 4.0 + pn * (double)m->p * 8.0 / 65000.0
                         : (m->n <= (long long)NT * (m->p <= 3 ? 20 : (m->p <= 7 ? 10 : 0)) ? 2.2 : now_rate * pn) + (kn->kind >= FMCMC_KERNEL_ADAPT ? 2.0 : 0.0);
    const void* kfn = fmh::k_wide(1, 2, kn->kind);       // (the long-data form: one chain per workgroup, every proposal kernel)
    const int rcl = try_long(kfn, lds, est_now, false);
    if (rcl != FMCMC_OK) return rcl;
  }
  // the operand-order copy of the observation slots beyond the registers (EXT / adaptive MFMA forms): 4 ng doubles per streamed
  // observation -- for p = 1 several times the size of X.  When the device cannot give that memory the call still runs: on
  // the general streamed kernel, which needs none
  if (!launched_long && mfma_ng && mfma_ext) {
    const int ns_all = (int)((m->n + NT - 1) / NT), next = ns_all - mfma_ext;
    double* mfs = nullptr;
    const size_t nd = (size_t)NW * (next > 0 ? next : 1) * mfma_ng * 64 * 4;   // (everything resident: one slot of stand-in, read and never used)
    if (hipMallocAsync((void**)&mfs, sizeof(double) * nd, stream) != hipSuccess) {
      (void)hipGetLastError();
      mfma_ng = 0; mfma_ext = 0; mfma_ad = 0; pipe_opt = 0; lat_normal = false;
    } else {
      mfs_guard.p = mfs; mfs_guard.s = stream;
      if (next > 0) hipLaunchKernelGGL(mfma_build_stream, dim3(512), dim3(256), 0, stream, m->X, m->y, (long long)m->n, m->p, mfma_ng, mfma_ext, next, mfs);
      else (void)hipMemsetAsync(mfs, 0, sizeof(double) * nd, stream);
      A.mf_stream = mfs; A.mf_next = next > 0 ? next : 0;
    }
  }
  if (launched_long) {
  } else
  if (pipe_opt || mfma_ng) {
    double* ws = nullptr;
    const double fill_df = (kn->kind == FMCMC_KERNEL_RAM) ? A.ram_df : (A.variate == 1 ? -1.0 : 0.0);
    // Step windows: the normal / uniform kernels with the library's own stream.  Window 0 is an ordinary launch of the call's
    // first n0 steps; every later window is a launch of w + 1 steps whose step 1 re-evaluates the state the window starts
    // from (bit for bit the f0 it replaces) and whose steps 2 .. w + 1 are the call's next w steps (SweepArgs.win_cont).
    // The stream of a window (rows of w + 1 steps) is filled right in front of it into ONE reused buffer of <= ~256 MiB
    // (it was nchains x nsteps x (kz + 1) doubles, and the reason for the 4 GiB limit; measured at C2's shape, 1.2e5 steps:
    // windows of 320 / 1344 / 8192 steps 2.17 / 2.08 / 2.05 us per step -- a window costs ~40 us of launches, refill of the
    // 80 operand registers and one extra evaluation, so the buffer is as large as is reasonable, not cache-sized).  The Philox counter
    // is the ABSOLUTE step, so the variates, and with them every bit of the output, do not depend on the cut
    // (windows begin behind a step = 1 mod 32: the accept bitmap's words then line up).
    // Round 5: kernel_adapt / kernel_ram too (R/kernel_adapt.R:118-133, R/kernel_ram.R:129-152 are ONE loop of any length).  What
    // depends on the step -- `i > 2`, the mean of this call's rows before the first adaptation, eta(i, k), `i %% freq` -- reads
    // the call's step (SweepArgs.step_off + the window's), the running sum of the rows travels from window to window
    // (SweepArgs.win_sum), everything else (Sigma / S, the running mean, abs_iter) is the state the windows hand on anyway.
    const bool windowed = A.rng_mode == FMCMC_RNG_PHILOX && !mirror &&
                          (kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE || kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM);
    long long win = run->nsteps;
    if (windowed) {
      const long long per_step = (long long)run->nchains * (A.kz + 1) * 8;
      win = ((256ll << 20) / (per_step > 0 ? per_step : 1)) & ~31ll;
      if (win < 32) win = 32;     // (the kernels take windows from 32 steps; a 512-step floor let the buffer grow with nchains without bound)
      if (K.window >= 32) win = (long long)K.window & ~31ll;      // (diagnosis / tests: a window length)
    }
    const long long n0 = (windowed && run->nsteps > win + 1) ? win + 1 : run->nsteps;   // steps of window 0
    if (A.rng_mode == FMCMC_RNG_PHILOX) {
      // materialise the canonical stream: [C][rows] log u, then [C][rows][kz] z; rows = a window's steps (or the whole call)
      const long long rows = (n0 < run->nsteps) ? win + 1 : run->nsteps;
      const size_t items = (size_t)run->nchains * (size_t)rows;
      e = hipMallocAsync((void**)&ws, sizeof(double) * items * (size_t)(A.kz + 1), stream);
      if (e != hipSuccess) { set_err("hipMallocAsync(rng stream) failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
      ws_guard.p = ws; ws_guard.s = stream;
    }
    auto fill_stream = [&](SweepArgs& W, long long step_base_eff) {   // the stream of launch W, rows = W.nsteps
      const size_t items = (size_t)W.nchains * (size_t)W.nsteps;
      hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                         (unsigned long long)run->seed, step_base_eff, (long long)run->chain_base,
                         (long long)W.nchains, (long long)W.nsteps, A.kz, fill_df, ws, ws + items);
      W.fed_logu = ws;
      W.fed_z = ws + items;
      W.rng_mode = FMCMC_RNG_FED;
    };
    auto launch_fast = [&](const SweepArgs& A) {   // (shadows the call's arguments: one launch of the call, or one step window of it)
      const long long pblk = (A.nchains + 3) / 4;
      if (mfma_ng) {
        // fp64-MFMA evaluation (mh_mfma.hpp), four chains per workgroup
        const int ns = (int)((m->n + NT - 1) / NT);   // observation slots of 512
        const int kv = (kn->kind == FMCMC_KERNEL_NORMAL) ? 1 : 2;
        // (offsets from the buffer bases stay 32 bits -- the cheaper form, see mh_sweep_mfma's BIG -- while the samples of all chains
        //  and the stream of this launch stay below 4 GiB)
        const bool big = (unsigned long long)A.nchains * kn->k * (unsigned long long)A.ldS * 8ull >= (1ull << 32) ||
                         (unsigned long long)A.nchains * (unsigned long long)A.nsteps * (unsigned long long)A.kz * 8ull >= (1ull << 32);
        if (mfma_ad) {
          // kernel_adapt / kernel_ram / mirror kernels: the adaptive owners between the barriers of the streamed evaluation (mh_mfma_ad.hpp)
          g_kernel = "mfma-adaptive";
          const bool ad_short = mfma_ext == 1;    // (short data: one resident slot)
          const int kx = (mfma_ad == 3) ? -2 : (mfma_ad == 2) ? -1 : ((kf != kn->k) ? 0 : (mfma_ng == 1 ? (kn->k == 5 ? 5 : 0) : (kn->k == 9 ? 9 : 0)));
          const bool bnd = mfma_ad == 1 && kn->kind == FMCMC_KERNEL_RAM && ram_bounded;
          e = launch_k(fmh::k_mfma_ad(kn->kind, mfma_ng, kx, bnd ? 1 : 0, ad_short ? 1 : 0), pblk, NT, mfma_ad_lds_bytes(mfma_ad == 2), stream, A);
        } else if (mfma_ext) {
          g_kernel = "mfma-streamed";
          e = launch_k(fmh::k_mfma_ext(kv, mfma_ng, mfma_ext, big ? 1 : 0), pblk, NT, mfma_lds_bytes(), stream, A);
        } else {
          g_kernel = "mfma";
          e = launch_k(fmh::k_mfma(kv, mfma_ng, ns, big ? 1 : 0), pblk, NT, mfma_lds_bytes(), stream, A);
        }
      } else if (lat_normal) {
        // the latency form of the normal / uniform kernels (mh_lat.hpp): A.spec_cw = 1 .. 3 chains per workgroup
        if (spec_logit) {
          g_kernel = A.spec_cw == 1 ? "lat-logit1" : A.spec_cw == 2 ? "lat-logit2" : A.spec_cw == 3 ? "lat-logit3" : "lat-logit4";
          e = launch_k(fmh::k_lat_logit(m->p, kn->kind), (A.nchains + A.spec_cw - 1) / A.spec_cw, NT, fmh::k_lat_logit_lds(), stream, A);
        } else {
        g_kernel = A.spec_cw == 1 ? "lat1" : A.spec_cw == 2 ? "lat2" : A.spec_cw == 3 ? "lat3" : "lat4";
        e = launch_k(fmh::k_lat(m->p, kn->kind), (A.nchains + A.spec_cw - 1) / A.spec_cw, NT, lat_lds_bytes(), stream, A);
        }
      } else {
        // the wave-specialised kernel (mh_spec.hpp): A.spec_cw chains per workgroup
        const long long sblk = (A.nchains + A.spec_cw - 1) / A.spec_cw;
        if (spec_logit) {
          g_kernel = A.spec_cw == 1 ? "spec-logit-lat1" : A.spec_cw == 2 ? "spec-logit-lat2" : A.spec_cw == 3 ? "spec-logit-lat3" : "spec-logit";
          e = launch_k((kn->kind == FMCMC_KERNEL_ADAPT && kn->freq > 1) ? fmh::k_spec_ring(m->p, 1) : fmh::k_spec_logit(m->p, kn->kind), sblk, SPEC_NT, fmh::k_spec_logit_lds(kn->kind >= FMCMC_KERNEL_ADAPT ? 1 : 0), stream, A);
        } else {
        g_kernel = A.spec_cw == 1 ? "spec-lat1" : A.spec_cw == 2 ? "spec-lat2" : A.spec_cw == 3 ? "spec-lat3" : "spec";
        e = launch_k((kn->kind == FMCMC_KERNEL_ADAPT && kn->freq > 1 && adapt_ring) ? fmh::k_spec_ring(m->p, 0) : fmh::k_spec(m->p, kn->kind), sblk, SPEC_NT, spec_lds_bytes(pipe_opt, kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM), stream, A);
        }
      }
    };   // launch_fast
    {
      SweepArgs W = A;
      W.nsteps = n0;
      W.bits_stride = (run->nsteps + 31) >> 5;
      const long long kept_all = A.S;
      long long* wcount = nullptr;                                          // accept counts of one continuation window
      double* wsum = nullptr;
      if (n0 < run->nsteps) {   // (window counts, and behind them kernel_adapt's running sums of the call's rows)
        const size_t nsum = (kn->kind == FMCMC_KERNEL_ADAPT) ? (size_t)run->nchains * (size_t)kf : 0;
        e = hipMallocAsync((void**)&wcount, sizeof(long long) * (size_t)run->nchains + sizeof(double) * nsum, stream);
        if (e != hipSuccess) { set_err("hipMallocAsync(window counts) failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
        wc_guard.p = wcount; wc_guard.s = stream;
        if (nsum) wsum = reinterpret_cast<double*>(wcount + run->nchains);
      }
      W.win_sum = wsum;
      if (A.rng_mode == FMCMC_RNG_PHILOX) fill_stream(W, (long long)run->step_base);
      launch_fast(W);
      for (long long s0 = n0; s0 < run->nsteps && e == hipSuccess; ) {     // continuation windows
        const long long w = (run->nsteps - s0 < win) ? run->nsteps - s0 : win;
        const long long rows_done = fmcmc_kept_rows(s0, run->burnin, run->thin);
        W = A;
        W.nsteps = w + 1;
        W.win_cont = 1;
        W.fresh = 0;                 // (kernel state: what the window before wrote back)
        W.win_sum = wsum;
        W.step_off = s0 - 1;
        W.burnin = (run->burnin - s0 + 1 > 1) ? run->burnin - s0 + 1 : 1;
        W.thin_ctr0 = (s0 > run->burnin) ? (int)((s0 - run->burnin) % run->thin) : 0;
        W.bits_stride = (run->nsteps + 31) >> 5;
        W.samples = A.samples + rows_done;
        if (A.logpost) W.logpost = A.logpost + rows_done;
        if (A.draws) W.draws = A.draws + rows_done;
        if (A.accept_bits) W.accept_bits = A.accept_bits + ((s0 - 1) >> 5);
        W.S = kept_all - rows_done;
        W.accept_count = wcount;
        fill_stream(W, (long long)run->step_base + s0 - 1);
        launch_fast(W);
        hipLaunchKernelGGL(add_counts_kernel, dim3((unsigned)((run->nchains + 255) / 256)), dim3(256), 0, stream, A.accept_count, wcount,
                           (long long)run->nchains);
        s0 += w;
      }
    }
  } else
  if (resident) { g_kernel = "resident"; e = launch_k(fmh::k_resident(res_p, kn->kind), nblk, NT, lds, stream, A); }
  else if (!force && m->family == FMCMC_FAM_LOGISTIC && cw <= 4 && lds + sizeof(double) * (LG_LDS_DOUBLES + LG_LDS_TAIL) <= 160 * 1024 &&
           (kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE || kn->kind == FMCMC_KERNEL_ADAPT ||
            kn->kind == FMCMC_KERNEL_RAM)) {
    // (round 4: kernel_adapt / kernel_ram too -- the workflow vignette's own model is a logistic regression under kernel_adapt;
    //  tools/option_audit.py found them on the all-family kernel at 3.9x the time per step of the normal kernels)
    // logistic-only instantiations: the g table in LDS; up to 28 / cw - 1 covariates their number is a compile-time constant
    // of the evaluation loop and the coefficients of the CW chains live in SGPRs (mh_common.hpp, logit_partials), beyond that
    // the run-time loop (logit_partials_any) -- still with the table in LDS, which is what the all-family kernel lacks
    g_kernel = "streamed-logistic";
    const int lkv = kn->kind;   // 1 .. 4
    lds += sizeof(double) * (LG_LDS_DOUBLES + LG_LDS_TAIL);   // the table staged behind the chain blocks (16-byte aligned), logit_shard's control words
    // Observation-sharded form (mh_common.hpp, logit_shard): 256 workgroups of two canonical lanes each evaluate ALL chains
    // of the launch, up to 256 x cw of them; more chains run as consecutive launches.  Cost model (us per step): the
    // chain-sharded loop costs ~(p + 12) instructions per observation and chain, 4.5 + n cw (p + 12) 1.35e-5 with the
    // coefficients in SGPRs (p <= 28 / cw - 1; 2.8e-5 on the run-time loop beyond that) -- its lookups scatter over the table:
    // LDS-bound -- but never less than one pass of the workgroup over the data set at ~90 GB/s; the sharded form ~10 us of
    // hand-overs + n (p + 10.3) 1.78e-5 per 512 chains of a launch.  Knob shard=1 forces it for every eligible shape (tests),
    // shard=0 disables it.
    bool lshard = false;
    // few chains: the long-data form (shard_long<LOGISTIC>) -- the sharded loop below keeps ONE thread per chain busy with its whole
    // slice (n = 1e5, 1 .. 64 chains: 31 .. 36 us per step), the chain-sharded one walks the data set in one workgroup
    if (m->p >= 1 && m->p <= 16) {
      const double w1 = (double)m->n * (double)(m->p + 12), stream1 = (double)m->n * (double)(m->p + 1) * 8.0 / 9.0e4;
      const double chain1 = 4.5 + ((w1 * 1.35e-5 > stream1) ? w1 * 1.35e-5 : stream1);
      const double shard1 = 10.3 + 1.78e-5 * (double)m->n * ((double)m->p + 10.3);
      const void* kfl = fmh::k_logit(1, 1, lkv);
      const int rcl = try_long(kfl, lds, (chain1 < shard1 || m->p > 16) ? chain1 : shard1, true);
      if (rcl != FMCMC_OK) return rcl;
    }
    if (launched_long) {
    } else {
    const long long nb_launch = 256;
    const long long ch_launch = (nblk > nb_launch) ? nb_launch * cw : (long long)run->nchains;
    const int nslots = (int)((m->n + NT - 1) / NT);
    // (not the bounded kernel_ram: its second evaluation of a step runs only in the workgroups where a proposal was reflected
    //  -- the grid-wide evaluation needs every workgroup in every hand-over)
    if (K.shard != 0 && m->p >= 1 && m->p <= 16 && ncu == 256 && m->n >= 2 * NT && m->n < (1ll << 28) &&
        !(kn->kind == FMCMC_KERNEL_RAM && ram_bounded)) {
      // (refitted to tools/dispatch_audit.py, profiles/r04_dispatch_audit.md: n = 2e3 .. 1e5, p = 2, 5, 8, 64 .. 4096 chains)
      const double w = (double)m->n * (double)(m->p + 12);
      const double stream_us = (double)m->n * ((double)m->p + 0.5) * 8.0 / 9.0e4;      // a workgroup's pass over the data set (X only: the term does not read y) at ~90 GB/s
      const double loop_us = w * cw * ((m->p <= 28 / cw - 1) ? 1.35e-5 : 2.8e-5);
      const double rounds = (double)((nblk + ncu - 1) / ncu), launches = (double)((run->nchains + ch_launch - 1) / ch_launch);
      const double est_chain = (4.5 + (loop_us > stream_us ? loop_us : stream_us)) * rounds;
      const double passes = (double)((ch_launch + NT - 1) / NT);                       // chains per thread of the sharded loop
      // (round 5: the issue-priority turns of logit_shard, and for the normal / uniform kernels mh_sweep_logit2 -- four chains per
      //  workgroup whatever cw says --: 5 .. 15 % off every row of profiles/r05_dispatch_audit_logistic.md)
      const bool shadow_ok = K.shadow != 0 && kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE && kn->scheme == FMCMC_SCHEME_JOINT && kf == kn->k;
      const double launches_s = shadow_ok ? (double)((run->nchains + 4 * nb_launch - 1) / (4 * nb_launch)) : launches;
      const double passes_s = shadow_ok ? (double)(((run->nchains < 4 * nb_launch ? run->nchains : 4 * nb_launch) + NT - 1) / NT) : passes;
      const double est_shard = ((shadow_ok ? 8.5 : 10.0) + 2.2 * (passes_s - 1.0) +
                                (shadow_ok ? 1.62e-5 : 1.72e-5) * (double)m->n * ((double)m->p + 10.3) * passes_s) * launches_s;
      lshard = K.shard == 1 || est_shard < 0.95 * est_chain;
    }
    const void* kfn = nullptr;
    if (lshard) kfn = fmh::k_logit(cw <= 2 ? cw : 4, 1, lkv);
    if (lshard) {
      int coop = 0, perCU = 0;
      (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
      e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess || !coop ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kfn, NT, lds) != hipSuccess || (long long)perCU * ncu < nb_launch) {
        if (A.debug & 256) fprintf(stderr, "fmcmc_amd: sharded logistic evaluation not launched: err=%d coop=%d perCU=%d lds=%zu\n", (int)e, coop, perCU, lds);
        lshard = false;
      }
      e = hipSuccess;
    }
    if (lshard) {
      double* shw = nullptr;
      // (+ 8 observations behind the last slice: the pipelined loop's scalar loads run up to three passes ahead without a clamp)
      // (mh_sweep_logit2 holds four chains per workgroup whatever cw says: tables for the larger of the two launch widths)
      const long long ch_shadow = (run->nchains < 4 * nb_launch) ? (long long)run->nchains : 4 * nb_launch;
      const long long ch_tab = (ch_shadow > ch_launch) ? ch_shadow : ch_launch;
      const size_t nxs = (size_t)nb_launch * nslots * 2 * m->p + 8 * (size_t)m->p, nth = ((size_t)kn->k * (ch_tab + SH_PAD) + 7) & ~(size_t)7,
                   npt = (size_t)(NT + SH_PAD) * ch_tab, nbar = 32 * 20 / 2;
      e = hipMallocAsync((void**)&shw, sizeof(double) * (nxs + nth + npt + nbar), stream);
      if (e != hipSuccess) { set_err("hipMallocAsync(sharded evaluation) failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
      shw_guard.p = shw; shw_guard.s = stream;
      double* thw = shw; double* ptw = thw + nth; unsigned* bar = (unsigned*)(ptw + npt); double* xs = ptw + npt + nbar;
      (void)hipMemsetAsync(xs + (size_t)nb_launch * nslots * 2 * m->p, 0, sizeof(double) * 8 * (size_t)m->p, stream);
      hipLaunchKernelGGL(logit_build_slices, dim3((unsigned)nb_launch), dim3(256), 0, stream, m->X, (long long)m->n, m->p, nslots, xs);
      A.shard = 2; A.sh_nslots = nslots; A.sh_xs = xs; A.sh_ys = nullptr; A.sh_th = thw; A.sh_part = ptw; A.sh_bar = bar;
      A.sh_t10 = (K.turn >= 0) ? K.turn : 1600700;   // (logit_shard's issue-priority turn: starts at 0.700 of the younger wave's passes, regulated towards a lead of 16 x 256 cycles; knob turn)
      g_kernel = "logistic-sharded";
      // Round 5: the canonical stream of the call materialised in front of the sweep (rng_fill_kernel), where it fits 1 GiB, instead
      // of being drawn inside the cooperative kernel: there the draws of a tile of steps -- Philox, AS241 with its ~50 constants
      // reloaded from scratch -- sit between two grid-wide hand-overs with 255 workgroups waiting (C5: 1.4 us of a 66 us step,
      // tools/bench_c5_fed.py).  The same variates, the same bits.
      SweepArgs A_own = A;
      const unsigned long long stream_bytes = (unsigned long long)run->nchains * (unsigned long long)run->nsteps * (unsigned long long)(A.kz + 1) * 8ull;
      if (A.rng_mode == FMCMC_RNG_PHILOX && stream_bytes <= (1ull << 30) &&
          (kn->kind >= FMCMC_KERNEL_ADAPT || kn->scheme == FMCMC_SCHEME_JOINT)) {
        double* wsl = nullptr;
        if (hipMallocAsync((void**)&wsl, (size_t)stream_bytes, stream) == hipSuccess) {
          ws_guard.p = wsl; ws_guard.s = stream;
          const size_t items = (size_t)run->nchains * (size_t)run->nsteps;
          const double fill_df = (kn->kind == FMCMC_KERNEL_RAM) ? A.ram_df : (A.variate == 1 ? -1.0 : 0.0);
          hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                             (unsigned long long)run->seed, (long long)run->step_base, (long long)run->chain_base,
                             (long long)run->nchains, (long long)run->nsteps, A.kz, fill_df, wsl, wsl + items);
          A.fed_logu = wsl; A.fed_z = wsl + items; A.rng_mode = FMCMC_RNG_FED;
        } else {
          (void)hipGetLastError();
        }
      }
      // (variates from a stream, the library's or the caller's: the instantiation without the generators in its body)
      const void* kfr = kfn;
      size_t lds_run = lds;
      long long ch_run = ch_launch;
      if (A.rng_mode == FMCMC_RNG_FED) {
        const void* kf2 = fmh::k_logit(cw <= 2 ? cw : 4, 2, lkv);
        if (kf2 && hipFuncSetAttribute(kf2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) kfr = kf2;
        else (void)hipGetLastError();
        // the normal / uniform proposal kernels, joint scheme, no fixed parameter: mh_sweep_logit2 (mh_logit2.hpp) -- four chains per
        // workgroup whatever cw says, the owners' work in the shadow of the hand-overs (knob shadow=0: off)
        // (kernel_adapt / kernel_ram with up to eight parameters, none fixed, no window / constraint / bound: the same sweep with the
        //  register owner of mh_spec.hpp, mh_sweep_logit2a)
        const bool adaptive3 = (kn->kind == FMCMC_KERNEL_ADAPT && !adapt_hist) || (kn->kind == FMCMC_KERNEL_RAM && !kn->constr && !ram_bounded);
        const void* kf3 = (K.shadow == 0 || kf != kn->k || A.kz != kn->k) ? nullptr
                        : (kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE ? (kn->scheme == FMCMC_SCHEME_JOINT ? fmh::k_logit2(lkv) : nullptr)
                           : ((adaptive3 && kn->k <= SPEC_KA && kn->k <= PIPE_KMAX) ? fmh::k_logit2a(lkv) : nullptr));
        if (kf3) {
          const size_t lds3 = (kn->kind <= FMCMC_KERNEL_NORMAL_REFLECTIVE) ? fmh::k_logit2_lds(kn->k) : fmh::k_logit2a_lds();
          int per3 = 0;
          if (hipFuncSetAttribute(kf3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) == hipSuccess &&
              hipOccupancyMaxActiveBlocksPerMultiprocessor(&per3, kf3, NT, lds3) == hipSuccess && (long long)per3 * ncu >= nb_launch) {
            kfr = kf3; lds_run = lds3; ch_run = ch_shadow;
            g_kernel = "logistic-shadow";
          } else {
            (void)hipGetLastError();
          }
        }
      }
      long long done = 0;
      for (; done < run->nchains && e == hipSuccess; done += ch_run) {   // (the slices and tables serve every launch)
        SweepArgs W = chain_window(A, done, (run->nchains - done < ch_run) ? run->nchains - done : ch_run, kf);
        W.bits_stride = (run->nsteps + 31) >> 5;       // (the owners of mh_spec.hpp address the accept bitmap through it)
        (void)hipMemsetAsync(bar, 0, sizeof(double) * nbar, stream);
        void* kargs[] = {(void*)&W};
        e = hipLaunchCooperativeKernel(kfr, dim3((unsigned)nb_launch), dim3(NT), kargs, (unsigned int)lds_run, stream);
      }
      if (e != hipSuccess && done > ch_run) {    // a LATER window failed: the chains of the earlier windows have run
        set_err("HIP launch of chain window %lld failed (%s): the state of the first %lld chains is already advanced, the results of this call are invalid",
                (long long)(done / ch_run), hipGetErrorString(e), (long long)(done - ch_run));
        (void)hipGetLastError();
        return FMCMC_ERR_DEVICE;
      }
      if (e != hipSuccess) {                         // the runtime refused the first cooperative launch: nothing ran
        (void)hipGetLastError();
        e = hipSuccess;
        lshard = false;
        g_kernel = "streamed-logistic";
        A = A_own;
        A.shard = 0; A.sh_xs = nullptr; A.sh_th = nullptr; A.sh_part = nullptr; A.sh_bar = nullptr;
      }
    }
    if (!lshard) {
      e = launch_k(fmh::k_logit(cw <= 2 ? cw : 4, 0, lkv), nblk, NT, lds, stream, A);
    }
    }   // (not the long-data form)
  }
  else if (m->family == FMCMC_FAM_GAUSSIAN_LINREG && m->p >= 16 && cw <= 2 &&
           (kn->kind == FMCMC_KERNEL_RAM || kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE)) {
    // wide linear models (config C4: k = 50): one family and one proposal kernel compiled in, which leaves the streamed
    // evaluation the registers for 4 observations x 8 columns in flight per thread (mh_common.hpp)
    const int kv = kn->kind;   // 1, 2 or 4
    // Observation-sharded evaluation: one cooperative launch when the call has 128 or 256 workgroups, consecutive launches
    // of 256 workgroups when it has a multiple of that (more than 512 chains per GPU at two chains per workgroup)
    const int nslots = (int)((m->n + NT - 1) / NT);
    // (a launch may hold workgroups WITHOUT chains -- they own canonical lanes like the others -- so any chain count works:
    //  up to 512 chains run as one launch of 256 workgroups, exactly 128 workgroups keep 4 lanes each when n allows)
    const long long nb_launch = (nblk == 128 && 4 * nslots <= SH_MAXO) ? 128 : 256;
    const long long ch_launch = (nblk > nb_launch) ? nb_launch * cw : (long long)run->nchains;
    Knobs Kw = K;
    if (wide_switched && K.shard != 0) Kw.shard = 1;
    const int lpw = wide_sharded_lanes(Kw, m, kn, run, ram_bounded, ncu, nb_launch, cw);
    bool shard = lpw > 0;
    // the sharded evaluation is its own instantiation (OPT = lanes per workgroup): sharing one with the streamed loop
    // cost 200-300 spilled registers in BOTH paths
    const void* kfn = nullptr;
    if (shard) kfn = fmh::k_wide(cw, lpw, kv);
    if (A.debug & 256) fprintf(stderr, "fmcmc_amd: wide path nblk=%lld lpw=%d nslots=%d p=%d bounded=%d shard=%d\n", nblk, lpw, nslots, m->p, (int)ram_bounded, (int)shard);
    double* shw = nullptr;
    g_kernel = shard ? "streamed-wide-sharded" : "streamed-wide";
    const size_t lds_plain = lds;   // what the chain-sharded kernel needs, should the sharded forms below not launch
    // the slice product on the matrix cores (shard_columns_mfma): the slice lives in LDS behind the chain blocks
    const int mf_spg = shard ? (nslots + 4 / lpw - 1) / (4 / lpw) : 0, nmt = (mf_spg + 3) / 4;
    const int mblk = shm_hdr(nmt) + nmt * ((m->p + 3) / 4) * 64;
    // (three M-tiles of which the third holds values 8, 9 only, at the width with a compile-time K-block count -- config C4:
    //  its 8 rows go through two 4x4x4 MFMAs per K-block instead of a 16x16x4 that is half padding; knob t10=0: off)
    //  (the form reads values 0 .. SHM_T10_FULL - 1 of every lane group without a mask: slots spg h + t <= nslots - 2 are full)
    const bool t10_full = shard && mf_spg * (4 / lpw - 1) + (SHM_T10_FULL - 1) <= nslots - 2;
    const int t10 = (nmt == 3 && mf_spg <= 10 && (m->p + 3) / 4 == 12 && t10_full && K.t10 != 0) ? 1 : 0;
    bool mfma_form = shard && shard_mfma_enabled(K) && m->p <= 4 * SHM_KBMAX && mf_spg <= SHM_T &&
                     lds + sizeof(double) * (size_t)(mblk + 1) <= 160 * 1024;
    if (mfma_form) lds += sizeof(double) * (size_t)(mblk + 1);
    if (shard && !mfma_form && lpw * nslots > SH_MAXO) shard = false;      // (more than 40 observations per slice: the matrix-core form or none)
    if (shard) g_kernel = mfma_form ? "streamed-wide-sharded-mfma" : "streamed-wide-sharded";
    // the dataflow form (mh_wide2.hpp): owner and evaluator waves decoupled, two chain groups half a step out of phase.
    // It pays where the owners have real work to hide -- kernel_ram: 35.8 -> 27.9 us per step at C4 -- and costs the normal
    // kernels 6 % (26.2 against 24.6: their owner phase is short and two of eight waves no longer evaluate).
    // Knob wide2=0 keeps the sequential form everywhere, wide2=1 takes the dataflow form for every eligible call (tests).
    bool wide2 = false;
    {
      const bool w2on = K.wide2 == 1 || (K.wide2 != 0 && kn->kind == FMCMC_KERNEL_RAM);
      wide2 = shard && mfma_form && w2on && lpw == 2 && cw == 2 && nb_launch == 256 && ncu == 256 &&
              !(kn->kind == FMCMC_KERNEL_RAM && kn->constr) && (kn->kind == FMCMC_KERNEL_RAM || kn->scheme == FMCMC_SCHEME_JOINT) &&
              nmt >= 1 && nmt <= 3 && run->nsteps < 100000000 &&
              sizeof(double) * wide2_lds_doubles(kn->k, kf, kn->kind, A.kz, mblk) <= 160 * 1024;
    }
    if (wide2) {
      kfn = fmh::k_wide2(kv, nmt);
      lds = sizeof(double) * wide2_lds_doubles(kn->k, kf, kn->kind, A.kz, mblk);
      g_kernel = "wide-dataflow";
      A.sh_ngrp = (K.groups == 4) ? 4 : 2;
      A.sh_tiles = (K.tiles == 0 || A.sh_ngrp == 4) ? 0 : 1;
    }
    if (shard) {
      int coop = 0, perCU = 0;
      (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
      if (lds > 48 * 1024) e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess || !coop ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kfn, NT, lds) != hipSuccess || (long long)perCU * ncu < nb_launch) {
        if (A.debug & 256) fprintf(stderr, "fmcmc_amd: sharded evaluation not launched: err=%d coop=%d perCU=%d lds=%zu\n", (int)e, coop, perCU, lds);
        shard = false;
        g_kernel = "streamed-wide";
        lds = lds_plain;
      }
      e = hipSuccess;
    }
    if (shard) {
      const size_t nxs = mfma_form ? 0 : (size_t)nb_launch * m->p * SH_MAXO, nys = mfma_form ? 0 : (size_t)nb_launch * SH_MAXO, nth = ((size_t)kn->k * (ch_launch + SH_PAD) + 7) & ~(size_t)7,   // (the partials behind it stay 64-byte aligned)
                   npt = (size_t)(NT + SH_PAD) * ch_launch, nbar = wide2 ? 8 * W2_BARW / 2 : 32 * 20 / 2;   // (barrier words counted in doubles)
      const size_t nmf = mfma_form ? (size_t)nb_launch * mblk : 0;
      e = hipMallocAsync((void**)&shw, sizeof(double) * (nxs + nys + nth + npt + nbar + nmf), stream);
      if (e != hipSuccess) { set_err("hipMallocAsync(sharded evaluation) failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
      shw_guard.p = shw; shw_guard.s = stream;
      double* xs = shw; double* ys = xs + nxs; double* thw = ys + nys; double* ptw = thw + nth;
      unsigned* bar = (unsigned*)(ptw + npt);
      if (!mfma_form)
        hipLaunchKernelGGL(shard_build_slices, dim3((unsigned)nb_launch), dim3(256), 0, stream, m->X, m->y, (long long)m->n, m->p, lpw, nslots, xs, ys);
      A.shard = lpw; A.sh_nslots = nslots; A.sh_xs = xs; A.sh_ys = ys; A.sh_th = thw; A.sh_part = ptw; A.sh_bar = bar;
      if (mfma_form) {
        double* mf = ptw + npt + nbar;
        hipLaunchKernelGGL(shard_build_mfma, dim3((unsigned)nb_launch), dim3(256), 0, stream, m->X, m->y, (long long)m->n, m->p, lpw, nslots,
                           nmt, t10, mf, mblk);
        A.sh_mfma = mf; A.sh_mblk = mblk; A.sh_nmt = nmt; A.sh_t10 = t10;
      }
      long long done = 0;
      for (; done < run->nchains && e == hipSuccess; done += ch_launch) {   // (the slices and tables serve every launch)
        SweepArgs W = chain_window(A, done, (run->nchains - done < ch_launch) ? run->nchains - done : ch_launch, kf);
        (void)hipMemsetAsync(bar, 0, sizeof(double) * nbar, stream);
        void* kargs[] = {(void*)&W};
        e = hipLaunchCooperativeKernel(kfn, dim3((unsigned)nb_launch), dim3(NT), kargs, (unsigned int)lds, stream);
      }
      if (e != hipSuccess && done > ch_launch) {    // a LATER window failed: the chains of the earlier windows have run
        set_err("HIP launch of chain window %lld failed (%s): the state of the first %lld chains is already advanced, the results of this call are invalid",
                (long long)(done / ch_launch), hipGetErrorString(e), (long long)(done - ch_launch));
        (void)hipGetLastError();
        return FMCMC_ERR_DEVICE;
      }
      if (e != hipSuccess && done <= ch_launch) {   // the runtime refused the first cooperative launch after all: nothing ran,
        (void)hipGetLastError();                    // take the chain-sharded kernel
        e = hipSuccess;
        shard = false;
        g_kernel = "streamed-wide";
        A.shard = 0; A.sh_xs = nullptr; A.sh_ys = nullptr; A.sh_th = nullptr; A.sh_part = nullptr; A.sh_bar = nullptr;
        A.sh_mfma = nullptr; A.sh_mblk = 0; A.sh_nmt = 0; A.sh_t10 = 0;
        lds = lds_plain;
      }
    }
    if (shard) {
    } else
    e = launch_k(fmh::k_wide(cw, 0, kv), nblk, NT, lds, stream, A);
  }
  else { g_kernel = "streamed"; e = launch_k(fmh::k_general(cw), nblk, NT, lds, stream, A); }
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { set_err("HIP launch failed: %s", hipGetErrorString(e)); return FMCMC_ERR_DEVICE; }
  // timing ablations and stamps (knob mode=<bits>: 8 stamps in the draws buffer, 32 / 64 / 128 / 1024 parts of a step left out)
  // produce INVALID samples: a call made with one of them says so in fmcmc_last_kernel(), so that its results cannot pass for
  // a product run's
  if (K.mode & (8 | 32 | 64 | 128 | 1024)) {
    static thread_local char kbuf[96];
    snprintf(kbuf, sizeof(kbuf), "invalid-results(mode=%d):%s", K.mode, g_kernel);
    g_kernel = kbuf;
  }
  return FMCMC_OK;
}

int fmcmc_rng_stream_dev(uint64_t seed, int64_t step_base, int64_t chain_base, int64_t nchains, int64_t nsteps,
                         int32_t kz, double student_df, double* logu, double* z, void* hip_stream) {
  if (!logu || !z || nchains < 1 || nsteps < 1 || kz < 1) { set_err("fmcmc_rng_stream_dev: bad argument"); return FMCMC_ERR_ARG; }
  const size_t items = (size_t)nchains * (size_t)nsteps;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     (unsigned long long)seed, (long long)step_base, (long long)chain_base, (long long)nchains,
                     (long long)nsteps, (int)kz, student_df, logu, z);
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
  // `fixed`, `lb`, `ub` (and scale / scheme_seq where they are checked) live on the device.  A caller that passes their
  // host copies (fmcmc_kernel.h_*) gets a call that only enqueues work; otherwise the few bytes are read back here, which
  // synchronises the stream.
  uint8_t fx[MAXK];
  double lb[MAXK], ub[MAXK], sc[MAXK];
  int32_t seq[MAXK];
  if (kn->k < 1 || kn->k > MAXK) { set_err("k=%d outside [1,%d]", kn->k, MAXK); return FMCMC_ERR_UNSUPPORTED; }
  hipStream_t stream = (hipStream_t)hip_stream;
  const bool unif = (kn->kind == FMCMC_KERNEL_UNIF || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE);
  const bool expl = (is_simple_kind(kn->kind) && kn->scheme == FMCMC_SCHEME_EXPLICIT && kn->scheme_seq &&
                     kn->scheme_len >= 1 && kn->scheme_len <= MAXK);
  const bool mirrored = kn->h_fixed && kn->h_lb && kn->h_ub && (!unif || kn->h_scale) && (!expl || kn->h_scheme_seq);
  if (mirrored) {
    memcpy(fx, kn->h_fixed, (size_t)kn->k);
    memcpy(lb, kn->h_lb, sizeof(double) * (size_t)kn->k);
    memcpy(ub, kn->h_ub, sizeof(double) * (size_t)kn->k);
    if (unif) memcpy(sc, kn->h_scale, sizeof(double) * (size_t)kn->k);
    if (expl) memcpy(seq, kn->h_scheme_seq, sizeof(int32_t) * (size_t)kn->scheme_len);
  } else if (hipMemcpyAsync(fx, kn->fixed, kn->k, hipMemcpyDeviceToHost, stream) != hipSuccess ||
             hipMemcpyAsync(lb, kn->lb, kn->k * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
             hipMemcpyAsync(ub, kn->ub, kn->k * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
             (unif && hipMemcpyAsync(sc, kn->scale, kn->k * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) ||
             (expl && hipMemcpyAsync(seq, kn->scheme_seq, kn->scheme_len * sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) ||
             hipStreamSynchronize(stream) != hipSuccess) {
    set_err("cannot read kernel parameters from device memory");
    return FMCMC_ERR_DEVICE;
  }
  fmcmc_kernel kh = *kn;
  kh.fixed = fx; kh.lb = lb; kh.ub = ub;
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
  if (out->ld_rows != 0 && out->ld_rows != S) { set_err("fmcmc_out.ld_rows is honoured by fmcmc_mcmc_run_dev only (host buffers are dense)"); return FMCMC_ERR_ARG; }
  dout.ld_rows = 0;
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
    UP(ds.obs_arate, st->fresh ? nullptr : st->obs_arate, sizeof(double) * (size_t)C * k);
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
    DOWN(st->obs_arate, ds.obs_arate, sizeof(double) * (size_t)C * k);
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
      // NaN log-posterior: the message of R/mcmc.R:759-765; the engine's own conditions by name
      const char* what = "fun(par) is undefined.";
      switch (out->status[c]) {
        case FMCMC_CHAIN_NAN_LOGPOST: what = "fun(par) is undefined (NaN)."; break;
        case FMCMC_CHAIN_NAN_RATIO: what = "fun(par) is undefined (f1 - f0 is NaN)."; break;
        case FMCMC_CHAIN_NOT_PD: what = "'Sigma' is not positive definite."; break;
        case FMCMC_CHAIN_BAD_WINDOW: what = "subscript out of bounds: the rows kernel_adapt(bw / freq) adapts on reach before the first row of this call."; break;
        case FMCMC_CHAIN_SYNC_TIMEOUT: what = "a grid-wide hand-over of the observation-sharded evaluation timed out; the results of this call are invalid (FMCMC_AMD_DEBUG=shard=0 selects the chain-sharded kernel)."; break;
        default: break;
      }
      // (R/mcmc.R:759-765 attaches the fun / lb / ub hint to a NaN log-posterior only)
      const bool nan_status = out->status[c] == FMCMC_CHAIN_NAN_LOGPOST || out->status[c] == FMCMC_CHAIN_NAN_RATIO;
      set_err("%s (chain %lld, status %d).%s This error ocurred during step i = %lld",
              what, (long long)(run->chain_base + c), out->status[c],
              nan_status ? " Check either -fun- or the -lb- and -ub- parameters." : "", (long long)out->status_step[c]);
      rc = FMCMC_ERR_CHAIN;
      break;
    }
done:
  for (void* p : allocs) hipFree(p);
  if (stream) hipStreamDestroy(stream);
  return rc;
}

}  // extern "C"
