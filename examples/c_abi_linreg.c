/* examples/c_abi_linreg.c -- the C-ABI of include/fmcmc_amd.h used from plain C: no Python, no torch, host pointers only.
 * This is the call an R `.Call` shim makes: fill the plain-old-data structs, call
 * fmcmc_mcmc_run_host(), read R-shaped column-major matrices back.
 *
 *   gcc -O2 -Iinclude examples/c_abi_linreg.c -o c_abi_linreg -Lfmcmc_amd/lib -lfmcmc_amd -Wl,-rpath,$PWD/fmcmc_amd/lib
 *   ./c_abi_linreg in.bin out.bin
 *
 * in.bin : int64 n, p, C, k, nsteps, burnin, thin, seed; double X[p][n], y[n], initial[C][k], scale[k]
 * out.bin: double samples[C][k][S], logpost[C][S]; int64 accept_count[C]        (S = fmcmc_kept_rows(...))
 * Model: Gaussian linear regression with intercept (README.md:128-139), kernel_normal(scale) (R/kernel_normal.R:26-82).
 * Exit code: 0 ok, 2 usage / IO, 3 the library returned an error (message on stderr; no GPU -> FMCMC_ERR_DEVICE).
 */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmcmc_amd.h"

static int rd(void* p, size_t sz, size_t cnt, FILE* f) { return fread(p, sz, cnt, f) == cnt; }

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  if (fmcmc_abi_version() != FMCMC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 3; }
  FILE* f = fopen(argv[1], "rb");
  int64_t h[8];
  if (!f || !rd(h, sizeof(int64_t), 8, f)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  const int64_t n = h[0], p = h[1], C = h[2], k = h[3], nsteps = h[4], burnin = h[5], thin = h[6];
  double* X = malloc(sizeof(double) * (size_t)(p * n)); double* y = malloc(sizeof(double) * (size_t)n);
  double* theta0 = malloc(sizeof(double) * (size_t)(C * k)); double* scale = malloc(sizeof(double) * (size_t)k);
  if (!rd(X, sizeof(double), (size_t)(p * n), f) || !rd(y, sizeof(double), (size_t)n, f) ||
      !rd(theta0, sizeof(double), (size_t)(C * k), f) || !rd(scale, sizeof(double), (size_t)k, f)) {
    fprintf(stderr, "short input\n"); return 2;
  }
  fclose(f);

  fmcmc_model m; memset(&m, 0, sizeof m);
  m.family = FMCMC_FAM_GAUSSIAN_LINREG; m.p = (int32_t)p; m.n = n; m.X = X; m.y = y; m.intercept = 1; m.guard = 1;

  double* mu = calloc((size_t)k, sizeof(double)); double* lb = malloc(sizeof(double) * (size_t)k);
  double* ub = malloc(sizeof(double) * (size_t)k); uint8_t* fixed = calloc((size_t)k, 1);
  for (int64_t j = 0; j < k; j++) { lb[j] = -DBL_MAX; ub[j] = DBL_MAX; }
  fmcmc_kernel kn; memset(&kn, 0, sizeof kn);
  kn.kind = FMCMC_KERNEL_NORMAL; kn.k = (int32_t)k; kn.mu = mu; kn.scale = scale; kn.lb = lb; kn.ub = ub; kn.fixed = fixed;
  kn.scheme = FMCMC_SCHEME_JOINT; kn.freq = 1; kn.until = 1.0 / 0.0; kn.eps = 1e-4; kn.arate = 0.234; kn.nadapt = 4;

  fmcmc_run r; memset(&r, 0, sizeof r);
  r.nchains = C; r.nsteps = nsteps; r.burnin = burnin; r.thin = thin; r.seed = (uint64_t)h[7]; r.rng_mode = FMCMC_RNG_PHILOX;

  if (fmcmc_validate(&m, &kn, &r) != FMCMC_OK) { fprintf(stderr, "fmcmc_validate: %s\n", fmcmc_last_error()); return 3; }
  const int64_t S = fmcmc_kept_rows(nsteps, burnin, thin);

  fmcmc_state st; memset(&st, 0, sizeof st);
  st.theta0 = theta0; st.f0 = calloc((size_t)C, sizeof(double)); st.abs_iter = calloc((size_t)C, sizeof(int64_t));
  st.fresh = 1;
  fmcmc_out out; memset(&out, 0, sizeof out);
  out.samples = malloc(sizeof(double) * (size_t)(C * k * S)); out.logpost = malloc(sizeof(double) * (size_t)(C * S));
  out.accept_count = calloc((size_t)C, sizeof(int64_t)); out.status = calloc((size_t)C, sizeof(int32_t));
  out.status_step = calloc((size_t)C, sizeof(int64_t)); out.status_theta = calloc((size_t)(C * k), sizeof(double));

  const int rc = fmcmc_mcmc_run_host(&m, &kn, &r, &st, &out, 0 /* device */);
  if (rc != FMCMC_OK) { fprintf(stderr, "fmcmc_mcmc_run_host: rc = %d: %s\n", rc, fmcmc_last_error()); return 3; }

  f = fopen(argv[2], "wb");
  if (!f) { fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
  fwrite(out.samples, sizeof(double), (size_t)(C * k * S), f);
  fwrite(out.logpost, sizeof(double), (size_t)(C * S), f);
  fwrite(out.accept_count, sizeof(int64_t), (size_t)C, f);
  fclose(f);
  /* chain 0: posterior means, the way summary(ans) would show them (samples[c][j][s] is column j of chain c's matrix) */
  printf("kernel %s, %lld kept rows per chain, chain 0 accepted %lld of %lld; means:", fmcmc_last_kernel(), (long long)S,
         (long long)out.accept_count[0], (long long)(nsteps - 1));
  for (int64_t j = 0; j < k; j++) {
    double s = 0; for (int64_t i = 0; i < S; i++) s += out.samples[(0 * k + j) * S + i];
    printf(" %.4f", s / (double)S);
  }
  printf("\n");
  return 0;
}
