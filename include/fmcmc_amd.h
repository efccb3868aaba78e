/* fmcmc_amd.h — C-ABI of the MI355X-native many-chain Metropolis-Hastings engine.
 *
 * fmcmc (USCbiostats/fmcmc v0.6-0) is interpreted R and has NO FFI for its hot path:
 * the boundary is the R closure protocol of the per-chain loop,
 *     draws[i,] <- kernel$proposal(environment())   R/mcmc.R:752
 *     logpost[i] <- f(theta1)                        R/mcmc.R:754
 *     klogratio <- kernel$logratio(environment())    R/mcmc.R:768
 * which cannot cross to a GPU.  The drop-in therefore replaces the WHOLE loop body of
 * MCMC_without_conv_checker (R/mcmc.R:720-838) for all chains of a call at once; the
 * entry points below are what an R `.Call` shim binds (INTEGRATION.md shows the stub).
 * Every struct is plain-old-data: plain pointers, sizes and scalars, no torch types.
 *
 * Two pointer domains, same structs:
 *   *_host entry points: every pointer is HOST memory; the library stages to/from HBM.
 *   *_dev  entry points: every pointer is DEVICE memory (hipMalloc / a torch tensor's
 *                        data_ptr); the call only enqueues work on `stream` when the kernel's
 *                        host mirrors (fmcmc_kernel.h_*) are set, else it reads a few bytes back first.
 *
 * Layouts (chosen so that one chain's block is exactly an R column-major matrix):
 *   X        [p][n]       column-major n x p design matrix (no intercept column)
 *   initial/theta0 [C][k] one row per chain (R: initial[c, ])
 *   samples  [C][k][S]    ans of chain c = column-major S x k matrix (R/mcmc.R:728,778)
 *   draws    [C][k][S]    proposals (R/mcmc.R:731,752)
 *   logpost  [C][S]       f(theta1) per kept row (R/mcmc.R:730,754)
 *   Sigma    [C][kf][kf]  row-major per chain; kernel_adapt: covariance (R/kernel_adapt.R:148),
 *                         kernel_ram: lower-triangular factor (R/kernel_ram.R:141)
 *   S = floor((nsteps - burnin) / thin) kept rows: 1-based rows r = burnin + j*thin
 *   (R/mcmc.R:786-813).
 */
#ifndef FMCMC_AMD_H
#define FMCMC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMCMC_ABI_VERSION 6
#define FMCMC_MAX_K 128 /* parameters per chain supported by the device kernels (R/kernel_ram.R:93-121, R/kernel_adapt.R:87-115: any k) */
/* Up to FMCMC_MAX_K_WAVE parameters a chain's rows live in the lanes of one wavefront and every kernel, scheme and option is
 * available; from there to FMCMC_MAX_K one workgroup serves a chain (mh_sweep_bigk): kernel_normal(_reflective) /
 * kernel_unif(_reflective) with scheme = "joint", kernel_adapt(bw = 0, freq = 1), kernel_ram -- anything else is refused above FMCMC_MAX_K_WAVE with FMCMC_ERR_UNSUPPORTED;
 * fmcmc_gelman_partial_dev (MFMA window tiles) takes p <= FMCMC_MAX_K_WAVE columns (above it the host side forms the same partial
 * sums itself, fmcmc_amd/convergence.py), fmcmc_gelman_finish any p <= FMCMC_MAX_K. */
#define FMCMC_MAX_K_WAVE 64

/* ---- log-posterior families: the `fun` argument of MCMC() (R/mcmc.R:327) ---------- */
enum {
  /* sum(dnorm(y - (b0 + X b), sd = sigma, log = TRUE)); theta = (b0?, b[p], sigma).
   * README.md:128-139 (guard=1: non-finite -> -Inf) and :356-361 (guard=0). */
  FMCMC_FAM_GAUSSIAN_LINREG = 1,
  /* sum(logp[y==1]) + sum(logq[y==0]) - sum(beta^2)/prior_div; theta = (b0?, b[p]).
   * vignettes/workflow-with-fmcmc.Rmd:35-41. */
  FMCMC_FAM_LOGISTIC = 2,
  /* sum(log(dnorm(D, mu, sigma))); theta = (mu, sigma); D passed as y, p = 0.
   * R/mcmc.R:141-144. */
  FMCMC_FAM_IID_NORMAL = 3
};

typedef struct fmcmc_model {
  int32_t family;     /* FMCMC_FAM_* */
  int32_t p;          /* covariate columns in X (0 allowed) */
  int64_t n;          /* observations */
  const double* X;    /* [p][n] column-major, may be NULL when p == 0 */
  const double* y;    /* [n] response / 0-1 outcome / data vector */
  int32_t intercept;  /* 1: theta[0] is an intercept (implicit column of ones) */
  int32_t guard;      /* 1: non-finite log-posterior is returned as -Inf (README.md:133-136) */
  double prior_div;   /* logistic: prior -sum(beta^2)/prior_div; 0 = flat */
} fmcmc_model;

/* ---- transition kernels: the `kernel` argument (R/kernel_*.R) --------------------- */
enum {
  FMCMC_KERNEL_NORMAL = 1,            /* R/kernel_normal.R:26-82   */
  FMCMC_KERNEL_NORMAL_REFLECTIVE = 2, /* R/kernel_normal.R:96-177  */
  FMCMC_KERNEL_ADAPT = 3,             /* R/kernel_adapt.R:54-202   */
  FMCMC_KERNEL_RAM = 4,               /* R/kernel_ram.R:65-181     */
  /* theta1[w] = theta0[w] + runif(min.[w], max.[w]) (R/kernel_unif.R:42-91, :96-170): the same proposal code as
   * the normal kernels with mu := min., scale := max. - min. and U(0,1) variates instead of N(0,1). */
  FMCMC_KERNEL_UNIF = 5,
  FMCMC_KERNEL_UNIF_REFLECTIVE = 6,
  /* mirror kernels (R/kernel_mirror.R:3-138, :140-283): theta1[w] ~ N(2 mu[w] - theta0[w], scale[w]^2) resp.
   * U(2 mu - theta0[w] -+ sqrt(3) scale); mu follows the running mean of the chain during warm-up and scale is rescaled
   * once, when abs_iter == nadapt, by tan(pi/2 obs_arate) / tan(pi/2 arate).  (The closure reads its ARGUMENT nadapt,
   * not the floor(seq(...)) vector stored next to it, so there is exactly one adaptation; the element-wise indexing of
   * kernel_umirror -- mu[a], scale[a] of the a-th updated parameter -- is kept as it is in the reference.) */
  FMCMC_KERNEL_NMIRROR = 7,
  FMCMC_KERNEL_UMIRROR = 8
};
/* Update schemes of the normal / uniform kernels, plan_update_sequence (R/kernel.R:66-133).  Row i of the plan is used
 * by loop step i (R/kernel_normal.R:67); the plan is built once per kernel object for the nsteps of its first call. */
enum {
  FMCMC_SCHEME_JOINT = 0,    /* all free parameters every step (:94-99) */
  FMCMC_SCHEME_ORDERED = 1,  /* one parameter per step, which(!fixed)[(i-1) mod kf] (:101-104) */
  FMCMC_SCHEME_RANDOM = 2,   /* one parameter per step, sample(which(!fixed), nsteps, TRUE)[i] (:106-113) */
  FMCMC_SCHEME_EXPLICIT = 3  /* one parameter per step, scheme_seq[(i-1) mod scheme_len] (:69-92) */
};

enum {
  FMCMC_RAM_QFUN_T_K = 0,    /* stats::rt(k, k), the default (R/kernel_ram.R:68) */
  FMCMC_RAM_QFUN_NORMAL = 1, /* stats::rnorm(k): the Gaussian proposal of Vihola (2012) */
  FMCMC_RAM_QFUN_T_DF = 2    /* stats::rt(k, ram_df) */
};

typedef struct fmcmc_kernel {
  int32_t kind;         /* FMCMC_KERNEL_* */
  int32_t k;            /* number of parameters (length of theta) */
  const double* mu;     /* [k] proposal mean (already recycled: R/kernel.R:3-17); unif kernels: min. */
  const double* scale;  /* [k] proposal sd (normal kernels); unif kernels: max. - min. */
  const double* lb;     /* [k] lower bounds, -DBL_MAX when unbounded (R/kernel.R:25-41) */
  const double* ub;     /* [k] upper bounds */
  const uint8_t* fixed; /* [k] 1 = parameter never updated */
  int32_t scheme;       /* FMCMC_SCHEME_* (normal / unif kernels) */
  int32_t freq;         /* adapt/ram: adaptation frequency (adapt every freq-th loop step, `!(env$i %% freq)`) */
  int32_t warmup;       /* adapt/ram */
  int32_t bw;           /* adapt: > 0 = windowed AM, Sigma = Sd * (cov(last bw - 1 rows) + eps * I) (R/kernel_adapt.R:123-125) */
  double until;         /* adapt/ram: stop adapting when abs_iter >= until (Inf allowed) */
  double eps;           /* adapt/ram: initial Sigma = eps * I */
  double arate;         /* ram: target acceptance rate */
  double Sd;            /* adapt: scaling of the windowed variant (the recursive path does not apply it, as in R) */
  const int32_t* scheme_seq; /* [scheme_len] FMCMC_SCHEME_EXPLICIT: 0-based parameter indices, a permutation of the
                              * free parameters (R/kernel.R:72-90); NULL otherwise */
  int32_t scheme_len;
  int32_t nadapt;       /* mirror kernels: abs_iter at which the scale is adapted (R/kernel_mirror.R:103,121) */
  const double* constr; /* ram: [kf][kf] mask multiplied element-wise into the updated factor
                         * (constr[which., which.], R/kernel_ram.R:149-150); NULL = none */
  /* v3, fmcmc_mcmc_run_dev only: HOST copies of the device arrays above that the launch geometry and the argument checks
   * need (fixed, lb, ub; scale for the uniform kernels; scheme_seq).  With all of them set the entry point only enqueues
   * work; with NULLs it reads the few bytes back from the device and synchronises the stream first. */
  const uint8_t* h_fixed;
  const double* h_lb;
  const double* h_ub;
  const double* h_scale;
  const int32_t* h_scheme_seq;
  /* ram: built-in alternatives to the defaults of R/kernel_ram.R:67-68 (user closures cannot cross a C boundary; these
   * are the families the arguments are used for).  All zero = the defaults. */
  int32_t ram_qfun;     /* FMCMC_RAM_QFUN_*: the variates U = qfun(k) of R/kernel_ram.R:124 */
  int32_t reserved;
  double ram_df;        /* FMCMC_RAM_QFUN_T_DF: degrees of freedom (> 0) */
  double ram_eta_exp;   /* eta(i, k) = min(1, k * i^(-ram_eta_exp)); 0 = the default 2/3 (R/kernel_ram.R:67) */
} fmcmc_kernel;

/* ---- one call of MCMC_without_conv_checker over all chains ------------------------ */
enum { FMCMC_RNG_PHILOX = 0, FMCMC_RNG_FED = 1 };

typedef struct fmcmc_run {
  int64_t nchains;      /* chains in THIS call (the shard) */
  int64_t nsteps;       /* R's nsteps: rows of ans before burn-in/thinning; loop runs i = 2..nsteps */
  int64_t burnin;
  int64_t thin;
  uint64_t seed;        /* Philox key */
  int64_t chain_base;   /* global id of local chain 0 (sharding never changes results) */
  int64_t step_base;    /* MH iterations already done by these chains in earlier calls/bulks */
  int32_t rng_mode;     /* FMCMC_RNG_* */
  int32_t reserved;
  /* FMCMC_RNG_FED: host-generated variates in R's own draw order (bit-exact fmcmc replay):
   * fed_logu [C][nsteps] (entry i-1 is R[i]); fed_z [C][nsteps][kz] proposal variates of step i
   * (kz = free parameters; normal/adapt: N(0,1); ram: Student-t).  NULL in PHILOX mode. */
  const double* fed_logu;
  const double* fed_z;
} fmcmc_run;

/* Per-chain state carried from call to call: chain restart by value (R/mcmc.R:344-422) plus
 * the kernel environments that persist across bulks (R/kernel.R:218-237). In/out. */
typedef struct fmcmc_state {
  double* theta0;      /* [C][k]  in: initial; out: last row of ans */
  double* f0;          /* [C]     out: log-posterior of theta0 */
  int64_t* abs_iter;   /* [C]     adapt/ram: kernel's abs_iter (0 for a fresh kernel) */
  double* Sigma;       /* [C][kf][kf] adapt/ram (kf = number of non-fixed parameters); NULL otherwise */
  double* mean_prev;   /* [C][kf] adapt: Mean_t_prev */
  int32_t* have_mean;  /* [C]     adapt: 0 while Mean_t_prev is NULL (R/kernel_adapt.R:130) */
  int32_t* nerrors;    /* [C]     ram: failed factor updates (R/kernel_ram.R:143) */
  int32_t fresh;       /* 1: kernel state is uninitialised; the engine sets Sigma = eps*I etc. */
  int32_t reserved;
  /* FMCMC_SCHEME_RANDOM: the plan of the kernel object, entry [c][i-1] = 0-based parameter updated by loop step i.
   * PHILOX mode: a pure function of (seed, global chain, i) -- identical in every call, as the reference reuses
   * the plan of the kernel's first call -- written here when non-NULL.  FED mode: read from here (the caller
   * replays R's sample()).  [C][nsteps] int32, or NULL. */
  int32_t* scheme_cols;
  /* mirror kernels: the adapted mean and scale of every chain, [C][k] each, and obs_arate, [C][k] (ABI 6; was [C]): NaN before
   * the one-off adaptation, the observed acceptance rate it used in all k entries after it, then -- through the rest of the
   * warm-up -- R's element-wise running mean of (ans[i-1, ] != ans[i-2, ]) (R/kernel_mirror.R:108-118,:246-253: the closure's
   * scalar turns into a k-vector there).  fresh == 1: mu / scale initialised from kernel->mu / kernel->scale. */
  double* mirror_mu;
  double* mirror_scale;
  double* obs_arate;
} fmcmc_state;

enum {
  FMCMC_CHAIN_OK = 0,
  FMCMC_CHAIN_NAN_LOGPOST = 1, /* fun(par) is undefined: R/mcmc.R:758-765 */
  FMCMC_CHAIN_NAN_RATIO = 2,   /* f1 - f0 is NaN (e.g. -Inf - -Inf): R's `if (NA)` error */
  FMCMC_CHAIN_NOT_PD = 3,      /* proposal covariance not positive definite (MASS::mvrnorm error) */
  /* kernel_adapt with bw > 0 or freq > 1: the rows ans[(i-bw+1):(i-1), ] / ans[(i-freq):(i-1), ] reach before the first row
   * of this call (R: "subscript out of bounds" / mixed subscripts, R/kernel_adapt.R:119-125,139-156) */
  FMCMC_CHAIN_BAD_WINDOW = 4,
  FMCMC_CHAIN_SYNC_TIMEOUT = 5 /* engine-internal: a grid-wide hand-over of the observation-sharded evaluation timed out */
};

typedef struct fmcmc_out {
  double* samples;        /* [C][k][S] */
  double* logpost;        /* [C][S] or NULL */
  double* draws;          /* [C][k][S] or NULL */
  int64_t* accept_count;  /* [C] accepted proposals in this call */
  uint32_t* accept_bits;  /* [C][ceil(nsteps/32)] bit (i-1) set iff loop step i accepted; or NULL */
  int32_t* status;        /* [C] FMCMC_CHAIN_* */
  int64_t* status_step;   /* [C] loop index i at which status was raised */
  double* status_theta;   /* [C][k] theta1 at that step */
  int64_t ld_rows;        /* v3, fmcmc_mcmc_run_dev only: row stride of samples / draws columns and of logpost rows; 0 = S.
                           * With ld_rows > S the kept rows of consecutive calls (the bulks of MCMC_with_conv_checker,
                           * R/mcmc.R:926-947) land in ONE preallocated [C][k][ld_rows] history: pass samples + rows so far. */
} fmcmc_out;

/* Return codes */
enum {
  FMCMC_OK = 0,
  FMCMC_ERR_ARG = 1,     /* invalid argument; message mirrors the reference's stop() text */
  FMCMC_ERR_DEVICE = 2,  /* HIP runtime failure / no device */
  FMCMC_ERR_CHAIN = 3,   /* at least one chain raised a FMCMC_CHAIN_* status */
  FMCMC_ERR_UNSUPPORTED = 4
};

int fmcmc_abi_version(void);
const char* fmcmc_last_error(void);
/* Diagnostic: which kernel variant the calling thread's last fmcmc_mcmc_run_* chose ("mfma", "spec", "resident",
 * "streamed", "streamed-logistic", "streamed-wide", "streamed-wide-sharded", ...).  Results never depend on it. */
const char* fmcmc_last_kernel(void);
int fmcmc_device_count(void);
int64_t fmcmc_kept_rows(int64_t nsteps, int64_t burnin, int64_t thin);

/* Validates a call exactly like R/mcmc.R:501-520 and the kernel initialisers
 * (R/kernel_normal.R:134-135, R/kernel.R:9,129-132); no GPU needed. */
int fmcmc_validate(const fmcmc_model* model, const fmcmc_kernel* kernel, const fmcmc_run* run);

/* The hot path. Replaces R/mcmc.R:720-838 x R/kernel_*.R for all chains of the call. */
int fmcmc_mcmc_run_dev(const fmcmc_model* model, const fmcmc_kernel* kernel,
                       const fmcmc_run* run, fmcmc_state* state, fmcmc_out* out,
                       void* hip_stream);
int fmcmc_mcmc_run_host(const fmcmc_model* model, const fmcmc_kernel* kernel,
                        const fmcmc_run* run, fmcmc_state* state, fmcmc_out* out,
                        int device);

/* Gelman-Rubin partial sums over local chains (R/convergence.R:191-246 -> coda::gelman.diag).
 * Window = kept rows [row0, row0+N) of each chain. partial has fmcmc_gelman_partial_len(p)
 * doubles: {m, sum xbar[p], sum xbar xbar^T[p*p], sum S_c[p*p], sum s2[p], sum s2^2[p],
 * sum s2*xbar[p], sum s2*xbar^2[p]}.  Partials of different GPUs add (one all-reduce). */
int64_t fmcmc_gelman_partial_len(int32_t p);
/* doubles of scratch (device) the partial kernel needs: per chain xbar[p] and S_c[p][p] */
int64_t fmcmc_gelman_work_len(int64_t nchains, int32_t p);
/* cols[p]: parameter indices to test (the free parameters, R/mcmc.R:950-968), device int32.
 * center[p] (device, may be NULL): xbar sums are accumulated relative to it (all ranks must
 * pass the same vector); it only limits cancellation, R-hat is shift-invariant. */
int fmcmc_gelman_partial_dev(const double* samples, int64_t nchains, int32_t k, int64_t S,
                             int64_t row0, int64_t N, const int32_t* cols, int32_t p,
                             const double* center, double* work, double* partial,
                             void* hip_stream);
/* Host finish on the (all-reduced) partial: psrf[p] point estimates, *mpsrf (NaN if p == 1).
 * Returns FMCMC_ERR_CHAIN when W is not positive definite (gelman.diag would fail). */
int fmcmc_gelman_finish(const double* partial, int32_t p, int64_t N, double* psrf,
                        double* mpsrf);

/* Materialises the canonical Philox stream of a call in device memory, in the FED layout of fmcmc_run:
 * logu[C][nsteps] (entry i-1 = log accept-uniform of loop step i), z[C][nsteps][kz] (N(0,1) when student_df == 0;
 * Student-t with student_df degrees of freedom when student_df > 0 -- kernel_ram: kf for the default qfun rt(k, k),
 * fmcmc_kernel.ram_df for FMCMC_RAM_QFUN_T_DF, 0 for FMCMC_RAM_QFUN_NORMAL (R/kernel_ram.R:68) -- ; U(0,1) of the uniform
 * kernels when student_df == -1).  A double since ABI 4 (it was an int32 and could not carry a fractional df).  A sweep run with
 * rng_mode = FMCMC_RNG_FED on these buffers is bit-identical to rng_mode = FMCMC_RNG_PHILOX; callers that
 * launch many sweeps can reuse the buffers instead of letting the library allocate them per call. */
int fmcmc_rng_stream_dev(uint64_t seed, int64_t step_base, int64_t chain_base, int64_t nchains, int64_t nsteps,
                         int32_t kz, double student_df, double* logu, double* z, void* hip_stream);

/* Diagnostic: evaluate the canonical math / RNG primitives on the device, element-wise
 * (which: 0 log, 1 exp, 2 log1p, 3 qnorm, 4 log accept-u, 5 normal, 6 student-t(df=x), 7 sqrt,
 * 8 reciprocal, 9 fused log1p(exp(x)) for x <= 0, 10 uniform variate). Used by tests to prove host/device bit-equality of include/fmh_*.h. */
int fmcmc_detmath_dev(int which, const double* x, double* out, int64_t n, uint64_t seed,
                      void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FMCMC_AMD_H */
