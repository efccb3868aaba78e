/* fmcmc_amd_shim.c -- the `.Call` side of the drop-in: what fmcmc (R) links to reach include/fmcmc_amd.h.
 *
 * Replaces, for all chains of a call, the body of MCMC_without_conv_checker (/root/reference R/mcmc.R:485-838): the R glue
 * in amd_hook.R builds four named lists (model, kernel, run, state), this file turns them into the POD structs of the
 * C-ABI, calls fmcmc_mcmc_run_host (blocking, host pointers) and returns R-owned result arrays laid out so that one
 * chain's block IS an R column-major matrix (samples [S x k x C], logpost [S x C], draws [S x k x C]).
 *
 * Rules kept here:
 *   * R owns every result buffer (allocVector + PROTECT); the library only fills them; inputs are borrowed for the call;
 *   * no R API is touched while the library runs; R_CheckUserInterrupt() before and after the blocking call (the unit an
 *     interrupt can cut is one call = one bulk of MCMC_with_conv_checker, R/mcmc.R:926-940);
 *   * errors: the library has released every HIP resource when it returns a code; the message is copied to a local buffer,
 *     the protect stack is unwound and only then Rf_error() longjmps (this file is C: no C++ frames are ever live);
 *   * the reference's stop() texts come from the library (fmcmc_last_error), the NaN log-posterior message of
 *     R/mcmc.R:759-765 is rebuilt here with the step and theta1 of the failing chain.
 *
 * R is not installed in this repository's image, so here the file is compiled and run against a stand-in for the handful of
 * R API calls it makes (tests/rapi_stub/, written from "Writing R Extensions"): tests/test_shim.py builds it with
 * -Wall -Werror -fsanitize=address,undefined, checks the argument errors against the reference's texts
 * (inst/tinytest/test-mcmc.R:3-23) and, on the GPU, drives C_fmcmc_amd_run from a C harness (tests/shim_harness.c) and
 * compares every returned array with the oracle bit for bit.  fmcmc_amd/_abi.py + engine.py mirror this file field by field.
 *
 * Build (inside the fmcmc package): src/Makevars from shim/Makevars.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Utils.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "fmcmc_amd.h"

/* ---- named-list access ------------------------------------------------------------------------------------------- */
static SEXP el(SEXP list, const char* name) {
  SEXP names = getAttrib(list, R_NamesSymbol);
  if (names == R_NilValue) return R_NilValue;
  for (R_xlen_t i = 0; i < XLENGTH(list); i++)
    if (!strcmp(CHAR(STRING_ELT(names, i)), name)) return VECTOR_ELT(list, i);
  return R_NilValue;
}
static int has(SEXP list, const char* name) { return el(list, name) != R_NilValue; }
static int el_int(SEXP list, const char* name, int dflt) {
  SEXP v = el(list, name);
  if (v == R_NilValue || XLENGTH(v) < 1) return dflt;
  if (TYPEOF(v) == INTSXP || TYPEOF(v) == LGLSXP) return INTEGER(v)[0];
  return (int)REAL(v)[0];
}
static double el_dbl(SEXP list, const char* name, double dflt) {
  SEXP v = el(list, name);
  if (v == R_NilValue || XLENGTH(v) < 1) return dflt;
  if (TYPEOF(v) == INTSXP || TYPEOF(v) == LGLSXP) return (double)INTEGER(v)[0];
  return REAL(v)[0];
}
/* numeric vector of exactly `n` doubles, or NULL when absent; the glue coerces with as.double() */
static double* el_real(SEXP list, const char* name, R_xlen_t n, int required) {
  SEXP v = el(list, name);
  if (v == R_NilValue) {
    if (required) error("fmcmc_amd shim: list element -%s- is missing", name);
    return NULL;
  }
  if (TYPEOF(v) != REALSXP) error("fmcmc_amd shim: -%s- must be double (use as.double)", name);
  if (n >= 0 && XLENGTH(v) != n) error("Incorrect length of -%s-.", name);   /* R/kernel.R:9 */
  return REAL(v);
}
static int* el_intvec(SEXP list, const char* name, R_xlen_t n, int required) {
  SEXP v = el(list, name);
  if (v == R_NilValue) {
    if (required) error("fmcmc_amd shim: list element -%s- is missing", name);
    return NULL;
  }
  if (TYPEOF(v) != INTSXP && TYPEOF(v) != LGLSXP) error("fmcmc_amd shim: -%s- must be integer", name);
  if (n >= 0 && XLENGTH(v) != n) error("Incorrect length of -%s-.", name);
  return INTEGER(v);
}
static void set_named(SEXP list, SEXP names, int i, const char* name, SEXP value) {
  SET_VECTOR_ELT(list, i, value);
  SET_STRING_ELT(names, i, mkChar(name));
}

/* ---- struct filling ---------------------------------------------------------------------------------------------- */
/* model = list(family, X (n x p double matrix or NULL), y, intercept, guard, prior_div) */
static void fill_model(SEXP model, fmcmc_model* m) {
  memset(m, 0, sizeof(*m));
  m->family = el_int(model, "family", 0);
  SEXP y = el(model, "y");
  if (y == R_NilValue || TYPEOF(y) != REALSXP) error("fmcmc_amd shim: model$y must be a double vector");
  m->n = (int64_t)XLENGTH(y);
  m->y = REAL(y);
  SEXP X = el(model, "X");
  if (X != R_NilValue && XLENGTH(X) > 0) {
    if (TYPEOF(X) != REALSXP) error("fmcmc_amd shim: model$X must be a double matrix");
    if (XLENGTH(X) % m->n) error("fmcmc_amd shim: model$X has %ld entries, not a multiple of length(y) = %ld",
                                 (long)XLENGTH(X), (long)m->n);
    m->p = (int32_t)(XLENGTH(X) / m->n);
    m->X = REAL(X);                       /* an R n x p matrix is [p][n] column-major: the ABI's layout, zero copy */
  }
  m->intercept = el_int(model, "intercept", 1);
  m->guard = el_int(model, "guard", 1);
  m->prior_div = el_dbl(model, "prior_div", 0.0);
}

/* kernel = list(kind, k, mu, scale, lb, ub, fixed (logical), scheme, freq, warmup, bw, until, eps, arate, Sd,
 *               scheme_seq (0-based integer) or NULL, nadapt, constr (kf x kf double, ROW-major = t() of R's) or NULL,
 *               ram_qfun, ram_df, ram_eta_exp (optional)) */
static void fill_kernel(SEXP kernel, fmcmc_kernel* k) {
  memset(k, 0, sizeof(*k));
  k->kind = el_int(kernel, "kind", 0);
  k->k = el_int(kernel, "k", 0);
  if (k->k < 1 || k->k > FMCMC_MAX_K) error("fmcmc_amd shim: number of parameters k = %d outside [1, %d]", k->k, FMCMC_MAX_K);
  const R_xlen_t K = k->k;
  k->mu = el_real(kernel, "mu", K, 1);
  k->scale = el_real(kernel, "scale", K, 1);
  k->lb = el_real(kernel, "lb", K, 1);
  k->ub = el_real(kernel, "ub", K, 1);
  const int* fx = el_intvec(kernel, "fixed", K, 1);
  uint8_t* f8 = (uint8_t*)R_alloc((size_t)K, 1);          /* freed by R at the end of the .Call */
  int kf = 0;
  for (R_xlen_t j = 0; j < K; j++) { f8[j] = fx[j] ? 1 : 0; kf += !f8[j]; }
  k->fixed = f8;
  k->scheme = el_int(kernel, "scheme", FMCMC_SCHEME_JOINT);
  k->freq = el_int(kernel, "freq", 1);
  k->warmup = el_int(kernel, "warmup", 0);
  k->bw = el_int(kernel, "bw", 0);
  k->until = el_dbl(kernel, "until", R_PosInf);
  k->eps = el_dbl(kernel, "eps", 1e-4);
  k->arate = el_dbl(kernel, "arate", 0.234);
  k->Sd = el_dbl(kernel, "Sd", 0.0);
  k->nadapt = el_int(kernel, "nadapt", 4);
  SEXP seq = el(kernel, "scheme_seq");
  if (seq != R_NilValue && XLENGTH(seq) > 0) {
    if (TYPEOF(seq) != INTSXP) error("fmcmc_amd shim: kernel$scheme_seq must be integer (0-based positions)");
    k->scheme_seq = (const int32_t*)INTEGER(seq);
    k->scheme_len = (int32_t)XLENGTH(seq);
  }
  k->constr = el_real(kernel, "constr", (R_xlen_t)kf * kf, 0);
  /* kernel_ram's qfun / eta families (amd_qfun_t / amd_qfun_normal / amd_eta_power of amd_hook.R); absent = defaults */
  k->ram_qfun = el_int(kernel, "ram_qfun", FMCMC_RAM_QFUN_T_K);
  k->ram_df = el_dbl(kernel, "ram_df", 0.0);
  k->ram_eta_exp = el_dbl(kernel, "ram_eta_exp", 0.0);
  /* host entry point: the arrays above ARE host memory, the mirrors are not needed */
}

/* run = list(nchains, nsteps, burnin, thin, seed (double, < 2^53), chain_base, step_base, rng_mode,
 *            fed_logu (nsteps x C) , fed_z (kz x nsteps x C)) */
/* a count that arrives as an R double: finite, integral and inside [lo, 2^53) (casting anything else is undefined behaviour) */
static int64_t el_count(SEXP list, const char* name, double dflt, double lo) {
  const double v = el_dbl(list, name, dflt);
  if (!(v >= lo && v < 9007199254740992.0) || v != floor(v))
    error("fmcmc_amd shim: -%s- must be a whole number in [%.0f, 2^53) (got %g)", name, lo, v);
  return (int64_t)v;
}
/* kz: proposal variates per step -- one for the single-parameter schemes of the simple kernels, else one per free parameter */
static R_xlen_t variates_per_step(const fmcmc_kernel* k) {
  R_xlen_t kf = 0;
  for (int j = 0; j < k->k; j++) kf += !k->fixed[j];
  const int simple = (k->kind == FMCMC_KERNEL_NORMAL || k->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE || k->kind == FMCMC_KERNEL_UNIF ||
                      k->kind == FMCMC_KERNEL_UNIF_REFLECTIVE || k->kind == FMCMC_KERNEL_NMIRROR || k->kind == FMCMC_KERNEL_UMIRROR);
  return (simple && k->scheme != FMCMC_SCHEME_JOINT) ? 1 : kf;
}
static void fill_run(SEXP run, const fmcmc_kernel* k, fmcmc_run* r) {
  memset(r, 0, sizeof(*r));
  /* (signed counts keep their sign: the range checks with the reference's messages are fmcmc_validate's) */
  r->nchains = (int64_t)el_count(run, "nchains", 1, -9007199254740991.0);
  r->nsteps = (int64_t)el_count(run, "nsteps", 0, -9007199254740991.0);
  r->burnin = (int64_t)el_count(run, "burnin", 0, -9007199254740991.0);
  r->thin = (int64_t)el_count(run, "thin", 1, -9007199254740991.0);
  r->seed = (uint64_t)el_count(run, "seed", 0, 0.0);          /* NA, negative or fractional seeds are refused, not cast */
  r->chain_base = el_count(run, "chain_base", 0, 0.0);
  r->step_base = el_count(run, "step_base", 0, 0.0);
  r->rng_mode = el_int(run, "rng_mode", FMCMC_RNG_PHILOX);
  if (r->rng_mode == FMCMC_RNG_FED && r->nchains > 0 && r->nsteps > 0) {
    /* el_count admits counts up to 2^53: form no product that can overflow (the extents an R array can have are INT_MAX) */
    if (r->nchains > INT_MAX || r->nsteps > INT_MAX || r->nchains > (int64_t)(PTRDIFF_MAX / 64) / r->nsteps)
      error("fmcmc_amd shim: nchains = %.0f x nsteps = %.0f exceed the extents of an R array", (double)r->nchains, (double)r->nsteps);
    const R_xlen_t cn = (R_xlen_t)r->nchains * (R_xlen_t)r->nsteps;
    r->fed_logu = el_real(run, "fed_logu", cn, 1);                               /* [C][nsteps] */
    r->fed_z = el_real(run, "fed_z", cn * variates_per_step(k), 1);              /* [C][nsteps][kz]: a short vector would be read out of bounds */
  }
}

/* ---- the call ---------------------------------------------------------------------------------------------------- */
/* .Call("C_fmcmc_amd_run", model, kernel, run, state)
 * state = list(theta0 (k x C: t(initial)), fresh, and -- when fresh = FALSE -- abs_iter (double[C]), Sigma (kf x kf x C, each
 *              slice the TRANSPOSE of R's matrix: the ABI is row-major), mean_prev (kf x C), have_mean (int[C]),
 *              nerrors (int[C]); scheme_cols (nsteps x C int) for a continued / fed "random" scheme;
 *              mirror_mu, mirror_scale, obs_arate (k x C) for the mirror kernels)
 * returns list(samples, logpost, draws, accept_count, status, status_step, status_theta, state = list(...same fields...),
 *              kernel_name) */
SEXP C_fmcmc_amd_run(SEXP model, SEXP kernel, SEXP run, SEXP state) {
  fmcmc_model m; fmcmc_kernel k; fmcmc_run r; fmcmc_state s; fmcmc_out o;
  fill_model(model, &m);
  fill_kernel(kernel, &k);
  fill_run(run, &k, &r);
  if (fmcmc_validate(&m, &k, &r) != FMCMC_OK) error("%s", fmcmc_last_error());   /* the reference's own stop() texts */
  const R_xlen_t C = (R_xlen_t)r.nchains, K = k.k, S = (R_xlen_t)fmcmc_kept_rows(r.nsteps, r.burnin, r.thin);
  /* allocMatrix / alloc3DArray take int extents: refuse what would be truncated instead of allocating something else */
  if (C > INT_MAX || S > INT_MAX || r.nsteps > INT_MAX)
    error("fmcmc_amd shim: %ld chains x %ld kept rows (%ld steps) exceed the extents of an R array", (long)C, (long)S, (long)r.nsteps);
  R_xlen_t kf = 0;
  for (R_xlen_t j = 0; j < K; j++) kf += !k.fixed[j];
  const int adaptive = (k.kind == FMCMC_KERNEL_ADAPT || k.kind == FMCMC_KERNEL_RAM);
  const int mirror = (k.kind == FMCMC_KERNEL_NMIRROR || k.kind == FMCMC_KERNEL_UMIRROR);
  const int fresh = el_int(state, "fresh", 1);
  int np = 0;                                                   /* protect counter */

  /* --- state: R-owned copies that the library updates in place and that go back to the glue */
  memset(&s, 0, sizeof(s));
  SEXP theta0 = PROTECT(allocMatrix(REALSXP, (int)K, (int)C)); np++;
  memcpy(REAL(theta0), el_real(state, "theta0", K * C, 1), sizeof(double) * (size_t)(K * C));
  SEXP f0 = PROTECT(allocVector(REALSXP, C)); np++;
  SEXP abs_iter = PROTECT(allocVector(REALSXP, C)); np++;      /* R has no int64: doubles on the R side */
  SEXP Sigma = PROTECT(alloc3DArray(REALSXP, (int)kf, (int)kf, (int)C)); np++;
  SEXP mean_prev = PROTECT(allocMatrix(REALSXP, (int)kf, (int)C)); np++;
  SEXP have_mean = PROTECT(allocVector(INTSXP, C)); np++;
  SEXP nerrors = PROTECT(allocVector(INTSXP, C)); np++;
  SEXP mirror_mu = PROTECT(allocMatrix(REALSXP, (int)K, (int)C)); np++;
  SEXP mirror_scale = PROTECT(allocMatrix(REALSXP, (int)K, (int)C)); np++;
  SEXP obs_arate = PROTECT(allocMatrix(REALSXP, (int)K, (int)C)); np++;
  int64_t* abs64 = (int64_t*)R_alloc((size_t)C, sizeof(int64_t));
  memset(REAL(f0), 0, sizeof(double) * (size_t)C);
  memset(REAL(Sigma), 0, sizeof(double) * (size_t)(kf * kf * C));
  memset(REAL(mean_prev), 0, sizeof(double) * (size_t)(kf * C));
  memset(INTEGER(have_mean), 0, sizeof(int) * (size_t)C);
  memset(INTEGER(nerrors), 0, sizeof(int) * (size_t)C);
  memset(REAL(mirror_mu), 0, sizeof(double) * (size_t)(K * C));
  memset(REAL(mirror_scale), 0, sizeof(double) * (size_t)(K * C));
  for (R_xlen_t c = 0; c < C; c++) abs64[c] = 0;
  for (R_xlen_t e = 0; e < K * C; e++) REAL(obs_arate)[e] = NA_REAL;
  if (!fresh) {
    const double* a = el_real(state, "abs_iter", C, adaptive || mirror);
    if (a) for (R_xlen_t c = 0; c < C; c++) abs64[c] = (int64_t)a[c];
    if (adaptive) {
      memcpy(REAL(Sigma), el_real(state, "Sigma", kf * kf * C, 1), sizeof(double) * (size_t)(kf * kf * C));
      if (k.kind == FMCMC_KERNEL_ADAPT) {
        memcpy(REAL(mean_prev), el_real(state, "mean_prev", kf * C, 1), sizeof(double) * (size_t)(kf * C));
        memcpy(INTEGER(have_mean), el_intvec(state, "have_mean", C, 1), sizeof(int) * (size_t)C);
      }
      if (has(state, "nerrors")) memcpy(INTEGER(nerrors), el_intvec(state, "nerrors", C, 1), sizeof(int) * (size_t)C);
    }
    if (mirror) {
      memcpy(REAL(mirror_mu), el_real(state, "mirror_mu", K * C, 1), sizeof(double) * (size_t)(K * C));
      memcpy(REAL(mirror_scale), el_real(state, "mirror_scale", K * C, 1), sizeof(double) * (size_t)(K * C));
      memcpy(REAL(obs_arate), el_real(state, "obs_arate", K * C, 1), sizeof(double) * (size_t)(K * C));
    }
  }
  s.theta0 = REAL(theta0); s.f0 = REAL(f0); s.abs_iter = abs64; s.Sigma = REAL(Sigma); s.mean_prev = REAL(mean_prev);
  s.have_mean = (int32_t*)INTEGER(have_mean); s.nerrors = (int32_t*)INTEGER(nerrors); s.fresh = fresh;
  s.mirror_mu = REAL(mirror_mu); s.mirror_scale = REAL(mirror_scale); s.obs_arate = REAL(obs_arate);
  /* update plan of scheme = "random" (update_sequence, R/kernel.R:106-113): in (continued / fed) or out (first call) */
  SEXP scheme_cols = R_NilValue;
  const int simple = (k.kind == FMCMC_KERNEL_NORMAL || k.kind == FMCMC_KERNEL_NORMAL_REFLECTIVE || k.kind == FMCMC_KERNEL_UNIF ||
                      k.kind == FMCMC_KERNEL_UNIF_REFLECTIVE || mirror);
  if (simple && k.scheme == FMCMC_SCHEME_RANDOM) {
    scheme_cols = PROTECT(allocMatrix(INTSXP, (int)r.nsteps, (int)C)); np++;
    const int* given = el_intvec(state, "scheme_cols", -1, 0);
    if (given) {
      const R_xlen_t rows = XLENGTH(el(state, "scheme_cols")) / C;      /* the plan has the rows of the kernel's FIRST call */
      if (rows < r.nsteps) error("subscript out of bounds");             /* R: update_sequence[env$i, ] */
      for (R_xlen_t c = 0; c < C; c++) memcpy(INTEGER(scheme_cols) + c * r.nsteps, given + c * rows, sizeof(int) * (size_t)r.nsteps);
    } else {
      if (r.rng_mode == FMCMC_RNG_FED) error("rng_mode = FED with scheme = 'random' needs state$scheme_cols (the plan R drew)");
      memset(INTEGER(scheme_cols), 0, sizeof(int) * (size_t)(r.nsteps * C));
    }
    s.scheme_cols = (int32_t*)INTEGER(scheme_cols);
  }

  /* --- results, owned by R */
  memset(&o, 0, sizeof(o));
  SEXP samples = PROTECT(alloc3DArray(REALSXP, (int)S, (int)K, (int)C)); np++;   /* [C][k][S]: C column-major S x k matrices */
  SEXP logpost = PROTECT(allocMatrix(REALSXP, (int)S, (int)C)); np++;
  SEXP draws = PROTECT(alloc3DArray(REALSXP, (int)S, (int)K, (int)C)); np++;
  SEXP accept_count = PROTECT(allocVector(REALSXP, C)); np++;
  SEXP status = PROTECT(allocVector(INTSXP, C)); np++;
  SEXP status_step = PROTECT(allocVector(REALSXP, C)); np++;
  SEXP status_theta = PROTECT(allocMatrix(REALSXP, (int)K, (int)C)); np++;
  int64_t* acc64 = (int64_t*)R_alloc((size_t)C, sizeof(int64_t));
  int64_t* step64 = (int64_t*)R_alloc((size_t)C, sizeof(int64_t));
  o.samples = REAL(samples); o.logpost = REAL(logpost); o.draws = REAL(draws);
  o.accept_count = acc64; o.accept_bits = NULL; o.status = (int32_t*)INTEGER(status); o.status_step = step64;
  o.status_theta = REAL(status_theta); o.ld_rows = 0;

  /* --- the blocking call: no R API from here until it returns */
  R_CheckUserInterrupt();
  const int device = el_int(run, "device", 0);
  const int rc = fmcmc_mcmc_run_host(&m, &k, &r, &s, &o, device);
  for (R_xlen_t c = 0; c < C; c++) {
    REAL(accept_count)[c] = (double)acc64[c];
    REAL(status_step)[c] = (double)step64[c];
    REAL(abs_iter)[c] = (double)abs64[c];
  }
  if (rc != FMCMC_OK) {
    char msg[2048];
    if (rc == FMCMC_ERR_CHAIN) {
      /* stop() of R/mcmc.R:759-765: "fun(par) is undefined (...) ... step i = ... theta1 = c(...)" for the first failing chain */
      R_xlen_t bad = 0;
      while (bad < C && INTEGER(status)[bad] == FMCMC_CHAIN_OK) bad++;
      if (bad == C) bad = 0;
      /* (built from the chain's status, not from fmcmc_last_error(): the library's text already ends in its own
       *  "This error ocurred during step i = N" and the sentence would appear twice) */
      const char* what = "fun(par) is undefined.";
      /* R/mcmc.R:759-765 attaches the fun / lb / ub hint to a NaN log-posterior only */
      const int nan_status = INTEGER(status)[bad] == FMCMC_CHAIN_NAN_LOGPOST || INTEGER(status)[bad] == FMCMC_CHAIN_NAN_RATIO;
      switch (INTEGER(status)[bad]) {
        case FMCMC_CHAIN_NAN_LOGPOST: what = "fun(par) is undefined (NaN)."; break;
        case FMCMC_CHAIN_NAN_RATIO: what = "fun(par) is undefined (f1 - f0 is NaN)."; break;
        case FMCMC_CHAIN_NOT_PD: what = "'Sigma' is not positive definite."; break;
        case FMCMC_CHAIN_BAD_WINDOW: what = "subscript out of bounds: the rows kernel_adapt(bw / freq) adapts on reach before the first row of this call."; break;
        case FMCMC_CHAIN_SYNC_TIMEOUT: what = "a grid-wide hand-over of the observation-sharded evaluation timed out; the results of this call are invalid."; break;
        default: break;
      }
      int off = snprintf(msg, sizeof msg, "%s%s This error ocurred during step i = %.0f "
                         "(chain %ld) and proposal parameters theta1 = c(", what, nan_status ? " Check either -fun- or the -lb- and -ub- parameters." : "",
                         REAL(status_step)[bad], (long)(r.chain_base + bad + 1));
      for (R_xlen_t j = 0; j < K && off < (int)sizeof msg - 40; j++)
        off += snprintf(msg + off, sizeof msg - (size_t)off, "%s%.4f", j ? ", " : "", REAL(status_theta)[bad * K + j]);
      snprintf(msg + off, sizeof msg - (size_t)off, ")");
    } else {
      snprintf(msg, sizeof msg, "%s", fmcmc_last_error());
    }
    UNPROTECT(np);                      /* every HIP resource was released by the library before it returned */
    error("%s", msg);                   /* longjmp; nothing of ours is live */
  }
  R_CheckUserInterrupt();

  /* --- return value */
  SEXP st = PROTECT(allocVector(VECSXP, 11)); np++;
  SEXP stn = PROTECT(allocVector(STRSXP, 11)); np++;
  set_named(st, stn, 0, "theta0", theta0);
  set_named(st, stn, 1, "f0", f0);
  set_named(st, stn, 2, "abs_iter", abs_iter);
  set_named(st, stn, 3, "Sigma", Sigma);
  set_named(st, stn, 4, "mean_prev", mean_prev);
  set_named(st, stn, 5, "have_mean", have_mean);
  set_named(st, stn, 6, "nerrors", nerrors);
  set_named(st, stn, 7, "scheme_cols", scheme_cols);
  set_named(st, stn, 8, "mirror_mu", mirror_mu);
  set_named(st, stn, 9, "mirror_scale", mirror_scale);
  set_named(st, stn, 10, "obs_arate", obs_arate);
  setAttrib(st, R_NamesSymbol, stn);
  SEXP ans = PROTECT(allocVector(VECSXP, 9)); np++;
  SEXP ansn = PROTECT(allocVector(STRSXP, 9)); np++;
  set_named(ans, ansn, 0, "samples", samples);
  set_named(ans, ansn, 1, "logpost", logpost);
  set_named(ans, ansn, 2, "draws", draws);
  set_named(ans, ansn, 3, "accept_count", accept_count);
  set_named(ans, ansn, 4, "status", status);
  set_named(ans, ansn, 5, "status_step", status_step);
  set_named(ans, ansn, 6, "status_theta", status_theta);
  set_named(ans, ansn, 7, "state", st);
  set_named(ans, ansn, 8, "kernel_name", mkString(fmcmc_last_kernel()));
  setAttrib(ans, R_NamesSymbol, ansn);
  UNPROTECT(np);
  return ans;
}

/* .Call("C_fmcmc_amd_validate", model, kernel, run): TRUE, or the reference's stop() text as an R error */
SEXP C_fmcmc_amd_validate(SEXP model, SEXP kernel, SEXP run) {
  fmcmc_model m; fmcmc_kernel k; fmcmc_run r;
  fill_model(model, &m);
  fill_kernel(kernel, &k);
  fill_run(run, &k, &r);
  if (fmcmc_validate(&m, &k, &r) != FMCMC_OK) error("%s", fmcmc_last_error());
  return ScalarLogical(1);
}

/* .Call("C_fmcmc_amd_info"): c(abi = , devices = ) -- amd_available() in the glue */
SEXP C_fmcmc_amd_info(void) {
  SEXP v = PROTECT(allocVector(INTSXP, 2));
  INTEGER(v)[0] = fmcmc_abi_version();
  INTEGER(v)[1] = fmcmc_device_count();
  UNPROTECT(1);
  return v;
}

static const R_CallMethodDef call_methods[] = {
  {"C_fmcmc_amd_run", (DL_FUNC)&C_fmcmc_amd_run, 4},
  {"C_fmcmc_amd_validate", (DL_FUNC)&C_fmcmc_amd_validate, 3},
  {"C_fmcmc_amd_info", (DL_FUNC)&C_fmcmc_amd_info, 0},
  {NULL, NULL, 0}};

void R_init_fmcmc(DllInfo* dll) {
  if (fmcmc_abi_version() != FMCMC_ABI_VERSION)
    error("libfmcmc_amd.so has ABI %d, this shim was built for %d", fmcmc_abi_version(), FMCMC_ABI_VERSION);
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
