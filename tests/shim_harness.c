/* tests/shim_harness.c -- drives shim/fmcmc_amd_shim.c the way R would: loads the "package" (R_init_fmcmc registers the
 * .Call table), looks the entry points up by name and arity, builds the named lists amd_hook.R builds, and runs each call
 * inside rapi_try (the stand-in for R's top-level error context, tests/rapi_stub/).  TEST INFRASTRUCTURE.
 *
 *   shim_harness errors                 argument errors, one "case|message" line each (no GPU needed: they are raised
 *                                       by the shim itself or by fmcmc_validate before anything touches the device)
 *   shim_harness run in.bin out.bin     one sweep through C_fmcmc_amd_run, then a SECOND call that continues from the
 *                                       returned state (fresh = FALSE): the state lists round-trip through the shim
 *     in.bin : int64 n, p, C, k, nsteps, burnin, thin, seed, kind, guard, nsteps2;
 *              double X[p][n], y[n], initial[C][k], scale[k]
 *     out.bin: double samples[C][k][S] | logpost[C][S] | draws[C][k][S] | accept_count[C] (as doubles) | theta0[C][k]
 *              of call 1, then the same five blocks of call 2
 *   exit code 0 ok, 2 usage / IO, 4 the call ended in an R error (message on stdout as "error|...")
 */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "Rinternals.h"
#include "R_ext/Rdynload.h"
#include "fmcmc_amd.h"

void R_init_fmcmc(DllInfo* dll);
typedef SEXP (*call3_t)(SEXP, SEXP, SEXP);
typedef SEXP (*call4_t)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*call0_t)(void);

static DllInfo dll;
struct args { DL_FUNC fn; int nargs; SEXP a[4]; };
static SEXP do_call(void* p) {
  struct args* a = (struct args*)p;
  switch (a->nargs) {
    case 0: return ((call0_t)a->fn)();
    case 3: return ((call3_t)a->fn)(a->a[0], a->a[1], a->a[2]);
    default: return ((call4_t)a->fn)(a->a[0], a->a[1], a->a[2], a->a[3]);
  }
}
/* .Call(name, ...): 0 and *out, or 1 with the message in rapi_last_error */
static int dot_call(const char* name, int nargs, SEXP a0, SEXP a1, SEXP a2, SEXP a3, SEXP* out) {
  struct args a;
  int registered = -1;
  a.fn = rapi_lookup(&dll, name, &registered);
  if (!a.fn) { fprintf(stderr, "routine %s is not registered\n", name); exit(2); }
  if (registered != nargs) { fprintf(stderr, "routine %s is registered with %d arguments, called with %d\n", name, registered, nargs); exit(2); }
  a.nargs = nargs; a.a[0] = a0; a.a[1] = a1; a.a[2] = a2; a.a[3] = a3;
  return rapi_try(do_call, &a, out);
}

static SEXP dbl1(double v) { return rapi_real(&v, 1); }
static SEXP int1(int v) { return rapi_int(&v, 1); }

/* the lists of amd_hook.R (kernel_spec / the model tag / the run list) for a Gaussian linear regression */
static SEXP model_list(const double* X, const double* y, int n, int p, int guard) {
  return rapi_list("family", int1(FMCMC_FAM_GAUSSIAN_LINREG), "X", p > 0 ? rapi_real(X, (R_xlen_t)n * p) : R_NilValue,
                   "y", rapi_real(y, n), "intercept", int1(1), "guard", int1(guard), "prior_div", dbl1(0.0), (const char*)NULL);
}
static SEXP kernel_list(int kind, int k, const double* scale, const double* lb, const double* ub, const int* fixed) {
  double mu[FMCMC_MAX_K];
  for (int j = 0; j < k; j++) mu[j] = 0.0;
  return rapi_list("kind", int1(kind), "k", int1(k), "mu", rapi_real(mu, k), "scale", rapi_real(scale, k), "lb", rapi_real(lb, k),
                   "ub", rapi_real(ub, k), "fixed", rapi_lgl(fixed, k), "scheme", int1(FMCMC_SCHEME_JOINT), "freq", int1(1),
                   "warmup", int1(0), "bw", int1(0), "until", dbl1(R_PosInf), "eps", dbl1(1e-4), "arate", dbl1(0.234),
                   "Sd", dbl1(0.0), "nadapt", int1(4), (const char*)NULL);
}
static SEXP run_list(double nchains, double nsteps, double burnin, double thin, double seed) {
  return rapi_list("nchains", dbl1(nchains), "nsteps", dbl1(nsteps), "burnin", dbl1(burnin), "thin", dbl1(thin), "seed", dbl1(seed),
                   "chain_base", dbl1(0), "step_base", dbl1(0), "rng_mode", int1(FMCMC_RNG_PHILOX), "device", int1(0), (const char*)NULL);
}

static void report(const char* name, int failed) { printf("%s|%s\n", name, failed ? rapi_last_error : "<no error>"); }

static int mode_errors(void) {
  /* the data of the reference's own error checks: one parameter, 100 steps (inst/tinytest/test-mcmc.R:3-23) -- here as an
   * iid_normal-sized linear model with no covariate (k = 2: intercept, sigma) */
  double y[8] = {0.1, -0.3, 0.7, 1.1, -0.9, 0.2, 0.4, -0.5};
  double scale[2] = {1, 1}, lb[2] = {-DBL_MAX, -DBL_MAX}, ub[2] = {DBL_MAX, DBL_MAX};
  int fixed[2] = {0, 0};
  SEXP model = model_list(NULL, y, 8, 0, 1);
  SEXP kern = kernel_list(FMCMC_KERNEL_NORMAL, 2, scale, lb, ub, fixed);
  SEXP out = NULL;
  report("info", dot_call("C_fmcmc_amd_info", 0, NULL, NULL, NULL, NULL, &out));
  if (out) printf("info_abi|%d\n", INTEGER(out)[0]);
  report("ok", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(1, 100, 0, 1, 1), NULL, &out));
  report("burnin", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(1, 100, 100, 1, 1), NULL, &out));
  report("thin_negative", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(1, 100, 0, -1, 1), NULL, &out));
  report("thin_nsteps", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(1, 100, 0, 100, 1), NULL, &out));
  report("nchains", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(0, 100, 0, 1, 1), NULL, &out));
  report("seed_negative", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(1, 100, 0, 1, -3), NULL, &out));
  report("seed_na", dot_call("C_fmcmc_amd_validate", 3, model, kern, run_list(1, 100, 0, 1, NA_REAL), NULL, &out));
  {   /* Incorrect length of -scale- (R/kernel.R:9) */
    double s3[3] = {1, 1, 1};
    SEXP k2 = kernel_list(FMCMC_KERNEL_NORMAL, 2, scale, lb, ub, fixed);
    for (R_xlen_t i = 0; i < XLENGTH(k2); i++)
      if (!strcmp(CHAR(STRING_ELT(getAttrib(k2, R_NamesSymbol), i)), "scale")) SET_VECTOR_ELT(k2, i, rapi_real(s3, 3));
    report("scale_length", dot_call("C_fmcmc_amd_validate", 3, model, k2, run_list(1, 100, 0, 1, 1), NULL, &out));
  }
  {   /* -ub- cannot be <= than -lb- (R/kernel_normal.R:134-135) */
    double lb2[2] = {0, 0}, ub2[2] = {1, 0};
    report("ub_lb", dot_call("C_fmcmc_amd_validate", 3, model, kernel_list(FMCMC_KERNEL_NORMAL_REFLECTIVE, 2, scale, lb2, ub2, fixed),
                             run_list(1, 100, 0, 1, 1), NULL, &out));
  }
  {   /* nothing to update (R/kernel.R:129-132) */
    int fx[2] = {1, 1};
    report("all_fixed", dot_call("C_fmcmc_amd_validate", 3, model, kernel_list(FMCMC_KERNEL_NORMAL, 2, scale, lb, ub, fx),
                                 run_list(1, 100, 0, 1, 1), NULL, &out));
  }
  {   /* a fed stream that is too short must be refused, not read out of bounds: kz = 2, 3 chains x 10 steps -> 60 variates */
    double lu[30], z[59];
    for (int i = 0; i < 30; i++) lu[i] = -1.0;
    for (int i = 0; i < 59; i++) z[i] = 0.0;
    SEXP r = rapi_list("nchains", dbl1(3), "nsteps", dbl1(10), "burnin", dbl1(0), "thin", dbl1(1), "seed", dbl1(1), "chain_base", dbl1(0),
                       "step_base", dbl1(0), "rng_mode", int1(FMCMC_RNG_FED), "fed_logu", rapi_real(lu, 30), "fed_z", rapi_real(z, 59),
                       (const char*)NULL);
    report("fed_z_short", dot_call("C_fmcmc_amd_validate", 3, model, kern, r, NULL, &out));
  }
  {   /* the run entry point raises the same texts before it touches the device */
    double th[2] = {0.0, 1.0};
    SEXP st = rapi_list("theta0", rapi_real(th, 2), "fresh", int1(1), (const char*)NULL);
    report("run_burnin", dot_call("C_fmcmc_amd_run", 4, model, kern, run_list(1, 100, 100, 1, 1), st, &out));
  }
  rapi_free_all();
  return 0;
}

static int rd(void* p, size_t sz, size_t cnt, FILE* f) { return fread(p, sz, cnt, f) == cnt; }
static void wr(FILE* f, const double* p, size_t cnt) { if (fwrite(p, sizeof(double), cnt, f) != cnt) { fprintf(stderr, "short write\n"); exit(2); } }

static int mode_run(const char* fin, const char* fout) {
  FILE* f = fopen(fin, "rb");
  int64_t h[11];
  if (!f || !rd(h, sizeof(int64_t), 11, f)) { fprintf(stderr, "cannot read %s\n", fin); return 2; }
  const int n = (int)h[0], p = (int)h[1], C = (int)h[2], k = (int)h[3], kind = (int)h[8], guard = (int)h[9];
  const double nsteps[2] = {(double)h[4], (double)h[10]}, burnin = (double)h[5], thin = (double)h[6], seed = (double)h[7];
  double* X = malloc(sizeof(double) * (size_t)p * n); double* y = malloc(sizeof(double) * (size_t)n);
  double* th = malloc(sizeof(double) * (size_t)C * k); double* scale = malloc(sizeof(double) * (size_t)k);
  if (!rd(X, sizeof(double), (size_t)p * n, f) || !rd(y, sizeof(double), (size_t)n, f) || !rd(th, sizeof(double), (size_t)C * k, f) ||
      !rd(scale, sizeof(double), (size_t)k, f)) { fprintf(stderr, "short input\n"); return 2; }
  fclose(f);
  double lb[FMCMC_MAX_K], ub[FMCMC_MAX_K];
  int fixed[FMCMC_MAX_K];
  for (int j = 0; j < k; j++) { lb[j] = -DBL_MAX; ub[j] = DBL_MAX; fixed[j] = 0; }
  SEXP model = model_list(X, y, n, p, guard);
  SEXP kern = kernel_list(kind, k, scale, lb, ub, fixed);
  /* state of a fresh kernel (kernel_state of amd_hook.R): theta0 = t(initial), k x C */
  SEXP state = rapi_list("theta0", rapi_real(th, (R_xlen_t)C * k), "fresh", int1(1), (const char*)NULL);
  FILE* g = fopen(fout, "wb");
  if (!g) { fprintf(stderr, "cannot write %s\n", fout); return 2; }
  double step_base = 0;
  for (int call = 0; call < 2; call++) {
    if (nsteps[call] <= 0) break;
    SEXP run = rapi_list("nchains", dbl1(C), "nsteps", dbl1(nsteps[call]), "burnin", dbl1(call ? 0 : burnin), "thin", dbl1(thin),
                         "seed", dbl1(seed), "chain_base", dbl1(0), "step_base", dbl1(step_base), "rng_mode", int1(FMCMC_RNG_PHILOX),
                         "device", int1(0), (const char*)NULL);
    SEXP ans = NULL;
    const int checks0 = rapi_interrupt_checks;
    if (dot_call("C_fmcmc_amd_run", 4, model, kern, run, state, &ans)) { printf("error|%s\n", rapi_last_error); fclose(g); return 4; }
    if (rapi_interrupt_checks - checks0 != 2) { fprintf(stderr, "R_CheckUserInterrupt ran %d times around the call\n", rapi_interrupt_checks - checks0); return 2; }
    SEXP samples = rapi_get(ans, "samples"), dim = getAttrib(samples, R_DimSymbol);
    const R_xlen_t S = INTEGER(dim)[0];
    if (INTEGER(dim)[1] != k || INTEGER(dim)[2] != C) { fprintf(stderr, "samples is not S x k x C\n"); return 2; }
    printf("call %d: kernel %s, %ld kept rows\n", call + 1, CHAR(STRING_ELT(rapi_get(ans, "kernel_name"), 0)), (long)S);
    wr(g, REAL(samples), (size_t)(S * k * C));
    wr(g, REAL(rapi_get(ans, "logpost")), (size_t)(S * C));
    wr(g, REAL(rapi_get(ans, "draws")), (size_t)(S * k * C));
    wr(g, REAL(rapi_get(ans, "accept_count")), (size_t)C);
    SEXP st = rapi_get(ans, "state");
    wr(g, REAL(rapi_get(st, "theta0")), (size_t)C * k);
    /* continue: the returned state lists go back in as they came out (kernel_write_back / kernel_state of amd_hook.R) */
    state = rapi_list("theta0", rapi_get(st, "theta0"), "fresh", int1(0), "abs_iter", rapi_get(st, "abs_iter"), "Sigma", rapi_get(st, "Sigma"),
                      "mean_prev", rapi_get(st, "mean_prev"), "have_mean", rapi_get(st, "have_mean"), "nerrors", rapi_get(st, "nerrors"),
                      (const char*)NULL);
    step_base += nsteps[call];
  }
  fclose(g);
  rapi_free_all();
  free(X); free(y); free(th); free(scale);
  return 0;
}

int main(int argc, char** argv) {
  memset(&dll, 0, sizeof dll);
  R_init_fmcmc(&dll);                       /* what R does when the package's shared object is loaded */
  if (argc == 2 && !strcmp(argv[1], "errors")) return mode_errors();
  if (argc == 4 && !strcmp(argv[1], "run")) return mode_run(argv[2], argv[3]);
  fprintf(stderr, "usage: %s errors | run in.bin out.bin\n", argv[0]);
  return 2;
}
