/* fmcmc_oracle.c — TEST INFRASTRUCTURE.  CPU restatement of fmcmc's hot path.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under fmcmc_amd/ links or imports it.
 *
 * WHAT IT RESTATES (reference = /root/reference, USCbiostats/fmcmc v0.6-0, pure R):
 *   R/mcmc.R:720-838        single-chain MH loop, burn-in, thinning       -> chain_run()
 *   R/kernel_normal.R:37-73, :109-165   kernel_normal(_reflective)        -> propose_normal()
 *   R/kernel.R:450-493      reflect_on_boundaries                         -> reflect()
 *   R/kernel_adapt.R:84-182 kernel_adapt (bw = 0 recursive path, freq=1)  -> propose_adapt()
 *   R/recursive.R:112-118, :124-128  cov_recursive / mean_recursive       -> cov_recursive1()
 *   R/kernel_ram.R:90-160   kernel_ram                                    -> propose_ram()
 *   R/convergence.R:191-246 convergence_gelman -> coda::gelman.diag       -> fmcmc_oracle_gelman()
 *   user `fun`: README.md:128-139,:356-361 (Gaussian linear regression),
 *               vignettes/workflow-with-fmcmc.Rmd:35-41 (logistic), R/mcmc.R:141-144 (iid normal)
 *
 * TWO MODES, one algorithm:
 *   rng_mode = ORACLE_RNG_RMT, math_mode = ORACLE_MATH_R
 *       fmcmc's bit-exact twin: R's Mersenne-Twister/inversion stream in R's serial draw order
 *       (accept uniforms first, chains one after another), libm, long-double sums, R's own
 *       operation order.  PINNED against the reference's printed outputs G1-G5
 *       (README.md:183-201, :315-339, :388-412, :245-246; R/mcmc_info.R:503-543) by
 *       tests/test_oracle_golden.py.
 *   rng_mode = ORACLE_RNG_PHILOX, math_mode = ORACLE_MATH_CANON
 *       the HIP engine's bit-exact twin: Philox stream (include/fmh_philox.h), deterministic
 *       math (include/fmh_detmath.h), the engine's canonical summation tree (512 lanes,
 *       xor-butterfly), Cholesky instead of eigen for mvrnorm, product-form factor update
 *       (S' = S chol(I +- p p^T), ram_factor_update_canon) for RAM.
 *   Third-party arithmetic not under /root/reference and how it is pinned: see r_rng.c
 *   (base R RNG: pinned by KATs + G1-G5), MASS::mvrnorm -> LAPACK dsyevr (parity UNPINNED at
 *   bit level, statistical only: G6), coda::gelman.diag mpsrf (pinned by G2/G3), psrf
 *   univariate branch (UNPINNED), Matrix::nearPD fallback (UNPINNED, never triggered).
 *   Also UNPINNED -- the reference prints no output of them, so they are restated from its source
 *   (and, for coda, from the published algorithms) and checked against independent numpy replays
 *   only: kernel_unif(_reflective) and the ordered / random / explicit update schemes
 *   (R/kernel_unif.R, R/kernel.R:60-113), kernel_nmirror / kernel_umirror (R/kernel_mirror.R),
 *   kernel_adapt(bw > 0 | freq > 1), kernel_ram(freq, constr) and the qfun / eta families
 *   (fmcmc_kernel.ram_*), convergence_geweke / heidel / auto (oracle.py: spectrum0.ar etc.).
 *   DESIGN.md section 3 carries the same list.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/fmcmc_amd.h"
#include "../include/fmh_detmath.h"
#include "../include/fmh_philox.h"
#include "r_rng.h"

#define ORACLE_RNG_PHILOX 0
#define ORACLE_RNG_RMT 1
#define ORACLE_MATH_CANON 0
#define ORACLE_MATH_R 1
#define ORACLE_LANES 512
#define MAXK FMCMC_MAX_K

typedef struct {
  int rng_mode, math_mode;
  r_rng* g;
  uint64_t seed;
  const double* lg_hs; /* canonical logistic: the data-only sums of logit_hs(), or NULL (then logpost_canon forms them) */
} ocfg;

/* ------------------------------------------------------------------------------------------
 * log-posterior families
 * ---------------------------------------------------------------------------------------- */

/* canonical 512-lane xor-butterfly tree (DESIGN.md "canonical reduction") */
static double tree512(double* acc) {
  for (int s = 1; s < ORACLE_LANES; s <<= 1)
    for (int l = 0; l < ORACLE_LANES; l += 2 * s) acc[l] = acc[l] + acc[l + s];
  return acc[0];
}

/* R's dnorm(x, 0, sigma, log = TRUE) (nmath/dnorm.c), x already y - mu */
static double r_dnorm0_log(double x, double sigma) {
  if (isnan(x) || isnan(sigma)) return x + sigma;
  if (sigma < 0) return NAN;
  if (!isfinite(sigma)) return -INFINITY;
  if (!isfinite(x) && 0.0 == x) return NAN;
  if (sigma == 0) return (x == 0) ? INFINITY : -INFINITY;
  x = x / sigma;
  if (!isfinite(x)) return -INFINITY;
  x = fabs(x);
  if (x >= 2 * sqrt(DBL_MAX)) return -INFINITY;
  return -(0.918938533204672741780329736406 + 0.5 * x * x + log(sigma));
}

/* R's dnorm(x, mu, sigma, log = FALSE) */
static double r_dnorm(double x, double mu, double sigma) {
  if (isnan(x) || isnan(mu) || isnan(sigma)) return x + mu + sigma;
  if (sigma < 0) return NAN;
  if (!isfinite(sigma)) return 0.0;
  if (!isfinite(x) && mu == x) return NAN;
  if (sigma == 0) return (x == mu) ? INFINITY : 0.0;
  x = (x - mu) / sigma;
  if (!isfinite(x)) return 0.0;
  x = fabs(x);
  if (x >= 2 * sqrt(DBL_MAX)) return 0.0;
  const double M_1_SQRT_2PI_ = 0.398942280401432677939946059934;
  if (x < 5) return M_1_SQRT_2PI_ * exp(-0.5 * x * x) / sigma;
  if (x > sqrt(-2 * M_LN2 * (DBL_MIN_EXP + 1 - DBL_MANT_DIG))) return 0.0;
  double x1 = ldexp(nearbyint(ldexp(x, 16)), -16);
  double x2 = x - x1;
  return M_1_SQRT_2PI_ / sigma * (exp(-0.5 * x1 * x1) * exp((-0.5 * x2 - x1) * x2));
}

static double logpost_R(const fmcmc_model* m, const double* th) {
  const int64_t n = m->n;
  const int p = m->p, ic = m->intercept ? 1 : 0;
  if (m->family == FMCMC_FAM_GAUSSIAN_LINREG) {
    /* README.md:130-131: sum(dnorm(y. - (p[1] + X.*p[2]), sd = p[3], log = TRUE)) */
    double sigma = th[ic + p];
    long double s = 0.0L;
    for (int64_t i = 0; i < n; i++) {
      double mu = ic ? th[0] : 0.0;
      for (int j = 0; j < p; j++) {
        double t = m->X[(int64_t)j * n + i] * th[ic + j];
        mu = (ic || j > 0) ? mu + t : t;
      }
      s += (long double)r_dnorm0_log(m->y[i] - mu, sigma);
    }
    double f = (double)s;
    if (m->guard && !isfinite(f)) return -INFINITY;
    return f;
  }
  if (m->family == FMCMC_FAM_IID_NORMAL) {
    /* R/mcmc.R:141-144: sum(log(dnorm(D, x[1], x[2]))) */
    long double s = 0.0L;
    for (int64_t i = 0; i < n; i++) s += (long double)log(r_dnorm(m->y[i], th[0], th[1]));
    double f = (double)s;
    if (m->guard && !isfinite(f)) return -INFINITY;
    return f;
  }
  if (m->family == FMCMC_FAM_LOGISTIC) {
    /* vignettes/workflow-with-fmcmc.Rmd:35-41 */
    long double s1 = 0.0L, s0 = 0.0L;
    for (int64_t i = 0; i < n; i++) {
      double eta = ic ? th[0] : 0.0;
      for (int j = 0; j < p; j++) eta += m->X[(int64_t)j * n + i] * th[ic + j];
      if (m->y[i] == 1.0) {
        double logp = (eta < 0) ? eta - log1p(exp(eta)) : -log1p(exp(-eta));
        s1 += (long double)logp;
      } else if (m->y[i] == 0.0) {
        double logq = (eta < 0) ? -log1p(exp(eta)) : -eta - log1p(exp(-eta));
        s0 += (long double)logq;
      }
    }
    double logl = (double)s1 + (double)s0;
    if (m->prior_div != 0.0) {
      long double ss = 0.0L;
      for (int j = 0; j < ic + p; j++) ss += (long double)(th[j] * th[j]);
      logl = logl - (double)ss / m->prior_div;
    }
    if (m->guard && !isfinite(logl)) return -INFINITY;
    return logl;
  }
  return NAN;
}

/* canonical logistic: hs[0] = sum_i w_i (intercept), hs[ic + j] = sum_i w_i x_ij with w_i = +1/2 (y_i != 0) or -1/2: every
 * product is exact, the sums run over the 512 canonical lanes in index order and their tree (the device: logit_hs_kernel) */
static void logit_hs(const fmcmc_model* m, double* hs) {
  const int64_t n = m->n;
  const int p = m->p, ic = m->intercept ? 1 : 0;
  double acc[ORACLE_LANES];
  for (int j = -ic; j < p; j++) {
    for (int l = 0; l < ORACLE_LANES; l++) acc[l] = 0.0;
    for (int64_t i = 0; i < n; i++) {
      const double w = (m->y[i] != 0.0) ? 0.5 : -0.5;
      const int l = (int)(i & (ORACLE_LANES - 1));
      acc[l] = acc[l] + (j < 0 ? w : w * m->X[(int64_t)j * n + i]);
    }
    hs[ic + j] = tree512(acc);
  }
}

static double logpost_canon(const fmcmc_model* m, const double* th, const double* hs) {
  const int64_t n = m->n;
  const int p = m->p, ic = m->intercept ? 1 : 0;
  double acc[ORACLE_LANES];
  for (int l = 0; l < ORACLE_LANES; l++) acc[l] = 0.0;
  if (m->family == FMCMC_FAM_GAUSSIAN_LINREG || m->family == FMCMC_FAM_IID_NORMAL) {
    int pp = (m->family == FMCMC_FAM_IID_NORMAL) ? 0 : p;
    int icc = (m->family == FMCMC_FAM_IID_NORMAL) ? 1 : ic;
    double sigma = th[icc + pp];
    for (int64_t i = 0; i < n; i++) {
      double mu = icc ? th[0] : 0.0;
      for (int j = 0; j < pp; j++) mu = fmh_fma(m->X[(int64_t)j * n + i], th[icc + j], mu);
      double r = m->y[i] - mu;
      int l = (int)(i & (ORACLE_LANES - 1));
      acc[l] = fmh_fma(r, r, acc[l]);
    }
    double ssr = tree512(acc);
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) {
      f = fmh_nan();
    } else if (sigma == 0.0) {
      f = -fmh_inf();
    } else {
      double t1 = fmh_log(sigma) + FMH_LN_SQRT_2PI;
      double q = (0.5 * ssr) / (sigma * sigma);
      f = -((double)n * t1) - q;
    }
    if (m->guard && !fmh_isfinite(f)) return -fmh_inf();
    return f;
  }
  if (m->family == FMCMC_FAM_LOGISTIC) {
    /* canonical form (include/fmh_detmath.h, fmh_logit_g): logl = sum_j b_j hs_j - sum_i g(|eta_i|), with eta formed from
     * the coefficients scaled by 64 (exact) so that the table's reduced argument falls out of it */
    double hs_own[MAXK], bsc[MAXK];
    if (!hs) { logit_hs(m, hs_own); hs = hs_own; }
    for (int j = 0; j < ic + p; j++) bsc[j] = th[j] * FMH_LG_SCALE;
    for (int64_t i = 0; i < n; i++) {
      double es = ic ? bsc[0] : 0.0;
      for (int j = 0; j < p; j++) es = fmh_fma(m->X[(int64_t)j * n + i], bsc[ic + j], es);
      int l = (int)(i & (ORACLE_LANES - 1));
      acc[l] = acc[l] + fmh_logit_g_scaled(fmh_abs(es));
    }
    double gs = tree512(acc);
    double lin = 0.0;
    for (int j = 0; j < ic + p; j++) lin = fmh_fma(th[j], hs[j], lin);
    double ll = lin - gs;
    if (m->prior_div != 0.0) {
      double ss = 0.0;
      for (int j = 0; j < ic + p; j++) ss = fmh_fma(th[j], th[j], ss);
      ll = ll - ss / m->prior_div;
    }
    if (m->guard && !fmh_isfinite(ll)) return -fmh_inf();
    return ll;
  }
  return fmh_nan();
}

static double logpost(const ocfg* cfg, const fmcmc_model* m, const double* th) {
  return cfg->math_mode == ORACLE_MATH_R ? logpost_R(m, th) : logpost_canon(m, th, cfg->lg_hs);
}

/* exported for tests */
double fmcmc_oracle_logpost(const fmcmc_model* m, const double* theta, int math_mode) {
  ocfg c; c.rng_mode = 0; c.math_mode = math_mode; c.g = NULL; c.seed = 0; c.lg_hs = NULL;
  return logpost(&c, m, theta);
}

/* ------------------------------------------------------------------------------------------
 * reflect_on_boundaries  (R/kernel.R:450-493)
 * ---------------------------------------------------------------------------------------- */
/* R's %% and %/% on doubles (arithmetic.c myfmod/myfloor), positive operands */
static double r_fmod(double x1, double x2) {
  if (x2 == 0.0) return NAN;
  if (fabs(x2) * DBL_EPSILON > 1 && isfinite(x1) && fabs(x1) <= fabs(x2))
    return (fabs(x1) == fabs(x2)) ? 0 : (((x1 < 0 && x2 > 0) || (x2 < 0 && x1 > 0)) ? x1 + x2 : x1);
  double q = x1 / x2;
  double tmp = x1 - floor(q) * x2;
  q = floor(tmp / x2);
  return tmp - q * x2;
}
static double r_intdiv(double x1, double x2) {
  double q = x1 / x2;
  if (x2 == 0.0 || fabs(q) * DBL_EPSILON > 1 || !isfinite(q)) return q;
  if (fabs(q) < 1) return (q < 0) ? -1 : (((x1 < 0 && x2 > 0) || (x1 > 0 && x2 < 0)) ? -1 : 0);
  long double tmp = (long double)x1 - floor(q) * (long double)x2;
  return (double)(floor(q) + floorl(tmp / x2));
}

static double reflect1(double x, double lb, double ub, int math_mode) {
  double d = ub - lb;
  if (x > ub) {
    double e = x - ub;
    if (math_mode == ORACLE_MATH_R) {
      double odd = r_fmod(r_intdiv(e, d), 2.0);
      double dm = r_fmod(e, d);
      return (lb + dm) * odd + (ub - dm) * (1 - odd);
    } else {
      double q = e / d, fq = __builtin_floor(q);
      double tmp = fmh_fma(-fq, d, e);
      double q2 = __builtin_floor(tmp / d);
      double dm = fmh_fma(-q2, d, tmp);
      double idiv = fq + q2;
      double odd = idiv - 2.0 * __builtin_floor(0.5 * idiv);
      return (odd != 0.0) ? (lb + dm) : (ub - dm);
    }
  }
  if (x < lb) {
    double e = lb - x;
    if (math_mode == ORACLE_MATH_R) {
      double odd = r_fmod(r_intdiv(e, d), 2.0);
      double dm = r_fmod(e, d);
      return (ub - dm) * odd + (lb + dm) * (1 - odd);
    } else {
      double q = e / d, fq = __builtin_floor(q);
      double tmp = fmh_fma(-fq, d, e);
      double q2 = __builtin_floor(tmp / d);
      double dm = fmh_fma(-q2, d, tmp);
      double idiv = fq + q2;
      double odd = idiv - 2.0 * __builtin_floor(0.5 * idiv);
      return (odd != 0.0) ? (ub - dm) : (lb + dm);
    }
  }
  return x;
}

void fmcmc_oracle_reflect(double* x, const double* lb, const double* ub, const int32_t* which,
                          int nwhich, int math_mode) {
  for (int a = 0; a < nwhich; a++) {
    int j = which[a];
    x[j] = reflect1(x[j], lb[j], ub[j], math_mode);
  }
}

/* ------------------------------------------------------------------------------------------
 * recursive mean / covariance  (R/recursive.R:124-128, :112-118), single new row
 * ---------------------------------------------------------------------------------------- */
void fmcmc_oracle_mean_recursive(const double* x, const double* mean_prev, double t, int k,
                                 double* mean_out) {
  for (int a = 0; a < k; a++) mean_out[a] = (mean_prev[a] * t + x[a]) / (t + 1);
}

/* cov [k][k] row-major, in/out. Ik is a k x k matrix (R passes diag(k)*eps of the kernel). */
void fmcmc_oracle_cov_recursive(const double* x, double* cov, const double* mean_prev,
                                const double* mean_t, double t, double eps, double Sd,
                                const double* Ik, int k) {
  double c1 = (t - 1) / t, c2 = Sd / t;
  for (int a = 0; a < k; a++)
    for (int b = 0; b < k; b++) {
      double inner = t * (mean_prev[a] * mean_prev[b]) - (t + 1) * (mean_t[a] * mean_t[b]) +
                     x[a] * x[b] + eps * Ik[a * k + b];
      cov[a * k + b] = c1 * cov[a * k + b] + c2 * inner;
    }
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers
 * ---------------------------------------------------------------------------------------- */
/* lower Cholesky, row-major; returns 0 ok, 1 not PD. canonical op order (fma). */
static int chol_lower_canon(const double* A, double* L, int k) {
  for (int a = 0; a < k * k; a++) L[a] = 0.0;
  for (int j = 0; j < k; j++) {
    double d = A[j * k + j];
    for (int b = 0; b < j; b++) d = fmh_fma(-L[j * k + b], L[j * k + b], d);
    if (!(d > 0.0) || !fmh_isfinite(d)) return 1;
    double ljj = fmh_sqrt(d);
    L[j * k + j] = ljj;
    for (int i = j + 1; i < k; i++) {
      double s = A[i * k + j];
      for (int b = 0; b < j; b++) s = fmh_fma(-L[i * k + b], L[j * k + b], s);
      L[i * k + j] = s / ljj;
    }
  }
  return 0;
}
/* Root-free factor Sigma = L D L^T (L unit lower, row-major; D the pivots), canonical op order -- the factor kernel_adapt's
 * canonical proposal draws through for k <= 64 parameters since round 5 (theta1 = theta0 + mu + L sqrt(D) z: the same law as
 * MASS::mvrnorm's eigen-factor and as the Cholesky factor it replaces, R/kernel_adapt.R:173-180).  W_ib = the numerator of
 * L_ib before its division (= L_ib D_b up to rounding) is what the sums subtract, so a column costs one division and no
 * square root; sqrt(D) is taken once, element-wise, at proposal time.  Returns 0 ok, 1 not positive definite. */
static int ldl_lower_canon(const double* A, double* L, double* D, int k) {
  static __thread double W[MAXK * MAXK];
  for (int a = 0; a < k * k; a++) L[a] = 0.0;
  for (int j = 0; j < k; j++) {
    double d = A[j * k + j];
    for (int b = 0; b < j; b++) d = fmh_fma(-L[j * k + b], W[j * k + b], d);
    if (!(d > 0.0) || !fmh_isfinite(d)) return 1;
    D[j] = d;
    L[j * k + j] = 1.0;
    for (int i = j + 1; i < k; i++) {
      double s = A[i * k + j];
      for (int b = 0; b < j; b++) s = fmh_fma(-L[i * k + b], W[j * k + b], s);
      W[i * k + j] = s;
      L[i * k + j] = s / d;
    }
  }
  return 0;
}
/* plain (no fma) lower Cholesky for the R-faithful mode */
static int chol_lower_plain(const double* A, double* L, int k) {
  for (int a = 0; a < k * k; a++) L[a] = 0.0;
  for (int j = 0; j < k; j++) {
    double d = A[j * k + j];
    for (int b = 0; b < j; b++) d -= L[j * k + b] * L[j * k + b];
    if (!(d > 0.0) || !isfinite(d)) return 1;
    double ljj = sqrt(d);
    L[j * k + j] = ljj;
    for (int i = j + 1; i < k; i++) {
      double s = A[i * k + j];
      for (int b = 0; b < j; b++) s -= L[i * k + b] * L[j * k + b];
      L[i * k + j] = s / ljj;
    }
  }
  return 0;
}

/* cyclic Jacobi eigen-decomposition of a symmetric matrix; values sorted decreasing,
 * V columns = eigenvectors (row-major V[a*k+j] = component a of vector j). */
static void jacobi_eig(const double* Ain, int k, double* ev, double* V) {
  double A[MAXK * MAXK];
  memcpy(A, Ain, sizeof(double) * k * k);
  for (int a = 0; a < k; a++)
    for (int b = 0; b < k; b++) V[a * k + b] = (a == b) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0;
    for (int a = 0; a < k; a++)
      for (int b = a + 1; b < k; b++) off += A[a * k + b] * A[a * k + b];
    if (off < 1e-300) break;
    for (int p = 0; p < k; p++)
      for (int q = p + 1; q < k; q++) {
        double apq = A[p * k + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * k + q] - A[p * k + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int r = 0; r < k; r++) {
          double arp = A[r * k + p], arq = A[r * k + q];
          A[r * k + p] = c * arp - s * arq;
          A[r * k + q] = s * arp + c * arq;
        }
        for (int r = 0; r < k; r++) {
          double apr = A[p * k + r], aqr = A[q * k + r];
          A[p * k + r] = c * apr - s * aqr;
          A[q * k + r] = s * apr + c * aqr;
        }
        for (int r = 0; r < k; r++) {
          double vrp = V[r * k + p], vrq = V[r * k + q];
          V[r * k + p] = c * vrp - s * vrq;
          V[r * k + q] = s * vrp + c * vrq;
        }
      }
  }
  for (int a = 0; a < k; a++) ev[a] = A[a * k + a];
  /* sort decreasing (selection) */
  for (int a = 0; a < k; a++) {
    int best = a;
    for (int b = a + 1; b < k; b++)
      if (ev[b] > ev[best]) best = b;
    if (best != a) {
      double t = ev[a]; ev[a] = ev[best]; ev[best] = t;
      for (int r = 0; r < k; r++) {
        double u = V[r * k + a]; V[r * k + a] = V[r * k + best]; V[r * k + best] = u;
      }
    }
  }
}

double fmcmc_oracle_top_eig_sym(const double* A, int k) {
  double ev[MAXK], V[MAXK * MAXK];
  jacobi_eig(A, k, ev, V);
  return ev[0];
}

/* ------------------------------------------------------------------------------------------
 * per-chain kernel state
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int k, kf;
  int which[MAXK];
  double Sigma[MAXK * MAXK]; /* [kf][kf] row-major */
  double mean_prev[MAXK];
  int have_mean;
  int64_t abs_iter;
  int nerrors;
  double run_sum[MAXK];        /* canonical: sum of ans rows 1..i-1 over free params */
  long double run_sum_ld[MAXK]; /* R mode */
  const int32_t* cols;         /* scheme = "random": plan of this chain, entry i-1 = column of loop step i */
  const double* hist;          /* adapt with bw > 0 / freq > 1: env$ans[, which.] of THIS call, row r at hist + (r-1)*kf */
  double mmu[MAXK], msc[MAXK]; /* mirror kernels: adapted mean / scale (all k parameters) */
  double obs_arate[MAXK];      /* R's obs_arate: a scalar (all k entries equal) after the one-off adaptation, element-wise afterwards */
  double prev_row[MAXK];       /* ans[i - 2, ] of this call (the row before theta0) */
  int64_t nzero;               /* rows r = 2..i-1 of this call with sum(diff(ans)[r-1, ]^2) == 0 */
} kstate;

/* draw helpers ---------------------------------------------------------------------------- */
static double draw_normal(const ocfg* cfg, uint32_t step, uint32_t chain, uint32_t j) {
  if (cfg->rng_mode == ORACLE_RNG_RMT) return r_norm_rand(cfg->g);
  return fmh_normal(cfg->seed, step, chain, j);
}
/* unif_rand() behind runif(k, min., max.) (R/kernel_unif.R:74; src/nmath/runif.c: a + (b - a) * u) */
static double draw_unif(const ocfg* cfg, uint32_t step, uint32_t chain, uint32_t j) {
  if (cfg->rng_mode == ORACLE_RNG_RMT) {
    double u;
    do { u = r_unif_rand(cfg->g); } while (u <= 0 || u >= 1);
    return u;
  }
  return fmh_unif(cfg->seed, step, chain, j);
}
static double draw_t(const ocfg* cfg, uint32_t step, uint32_t chain, uint32_t j, double df) {
  if (cfg->rng_mode == ORACLE_RNG_RMT) return r_rt(cfg->g, df);
  return fmh_student_t(cfg->seed, step, chain, j, df);
}

/* kernel_normal / kernel_normal_reflective (R/kernel_normal.R:65-72, :146-164) and kernel_unif / kernel_unif_reflective
 * (R/kernel_unif.R:70-76, :150-166; there mu = min., scale = max. - min. and the variate is unif_rand) */
static void propose_normal(const ocfg* cfg, const fmcmc_kernel* kn, const kstate* ks, int64_t i,
                           uint32_t step, uint32_t chain, const double* theta0, double* theta1) {
  for (int a = 0; a < kn->k; a++) theta1[a] = theta0[a];
  int upd[MAXK], nupd = 0;
  if (kn->scheme == FMCMC_SCHEME_ORDERED) {
    /* R/kernel.R:101-104: row r of the plan updates which(!fixed)[(r-1) mod kf] */
    upd[nupd++] = ks->which[(int)((i - 1) % ks->kf)];
  } else if (kn->scheme == FMCMC_SCHEME_EXPLICIT) {
    /* R/kernel.R:91-92: update_sequence[cbind(1:nsteps, scheme)] recycles scheme along the rows */
    upd[nupd++] = kn->scheme_seq[(int)((i - 1) % kn->scheme_len)];
  } else if (kn->scheme == FMCMC_SCHEME_RANDOM) {
    upd[nupd++] = ks->cols[i - 1]; /* R/kernel.R:106-113 */
  } else {
    for (int a = 0; a < ks->kf; a++) upd[nupd++] = ks->which[a];
  }
  const int unif = (kn->kind == FMCMC_KERNEL_UNIF || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE);
  for (int a = 0; a < nupd; a++) {
    int j = upd[a];
    /* src/nmath/rnorm.c, runif.c: sd == 0 (a == b) returns the mean (a) WITHOUT consuming a variate */
    const int skip = (cfg->rng_mode == ORACLE_RNG_RMT && kn->scale[j] == 0.0);
    double z = skip ? 0.0 : (unif ? draw_unif(cfg, step, chain, (uint32_t)a) : draw_normal(cfg, step, chain, (uint32_t)a));
    theta1[j] = theta1[j] + (kn->mu[j] + kn->scale[j] * z);
  }
  if (kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE)
    for (int a = 0; a < nupd; a++) {
      int j = upd[a];
      theta1[j] = reflect1(theta1[j], kn->lb[j], kn->ub[j], cfg->math_mode);
    }
}

/* kernel_adapt proposal (R/kernel_adapt.R:117-180), bw = 0, freq = 1. Returns chain status. */
/* kernel_nmirror / kernel_umirror proposal (R/kernel_mirror.R:66-131, :203-262).  `i` is the loop index, ans[i-1, ] ==
 * theta0; the plan column logic is the one of propose_normal. */
static void propose_mirror(const ocfg* cfg, const fmcmc_kernel* kn, kstate* ks, int64_t i, uint32_t step,
                           uint32_t chain, const double* theta0, double* theta1) {
  const int k = kn->k;
  const int64_t a_it = ks->abs_iter;
  if (a_it >= 1 && a_it <= kn->warmup) /* mu <<- mean_recursive(ans[i-1, ], mu, abs_iter) :92-100 */
    for (int a = 0; a < k; a++) ks->mmu[a] = (ks->mmu[a] * (double)a_it + theta0[a]) / ((double)a_it + 1);
  if (a_it == kn->nadapt) {
    /* obs_arate <<- 1 - mean(rowSums(diff(ans[1:(i-1), ])^2) == 0) :103-107 (mean of zero rows is NaN);
     * scale <<- scale * tan(pi/2 obs_arate) / tan(pi/2 arate) :121-127 */
    const double oa = 1.0 - (double)ks->nzero / (double)(i - 2);
    for (int a = 0; a < k; a++) ks->obs_arate[a] = oa;
    double num, den;
    if (cfg->math_mode == ORACLE_MATH_R) {
      num = tan(M_PI / 2.0 * oa);
      den = tan(M_PI / 2.0 * kn->arate);
    } else {
      num = fmh_tan_0_halfpi(1.5707963267948966 * oa);
      den = fmh_tan_0_halfpi(1.5707963267948966 * kn->arate);
    }
    for (int a = 0; a < k; a++) ks->msc[a] = ks->msc[a] * num / den;
  } else if (a_it > kn->nadapt && a_it <= kn->warmup) {
    /* obs_arate <<- mean_recursive(as.double(ans[i-1, ] != ans[i-2, ]), obs_arate, abs_iter) :108-118, :246-253 -- element-wise over the
     * k parameters: (obs t + X_t) / (t + 1) (R/recursive.R:124-127).  At the first proposal of a call (i = 2) ans[i-2, ] is ans[0, ]:
     * zero rows, the comparison and with it obs_arate become numeric(0) in R and stay so; here: NaN in every entry (NaN t + X = NaN) */
    for (int a = 0; a < k; a++)
      ks->obs_arate[a] = (i < 3) ? NAN : (ks->obs_arate[a] * (double)a_it + ((theta0[a] != ks->prev_row[a]) ? 1.0 : 0.0)) / ((double)a_it + 1);
  }
  for (int a = 0; a < k; a++) theta1[a] = theta0[a];
  int upd[MAXK], nupd = 0;
  if (kn->scheme == FMCMC_SCHEME_ORDERED) upd[nupd++] = ks->which[(int)((i - 1) % ks->kf)];
  else if (kn->scheme == FMCMC_SCHEME_EXPLICIT) upd[nupd++] = kn->scheme_seq[(int)((i - 1) % kn->scheme_len)];
  else if (kn->scheme == FMCMC_SCHEME_RANDOM) upd[nupd++] = ks->cols[i - 1];
  else for (int a = 0; a < ks->kf; a++) upd[nupd++] = ks->which[a];
  const int rmt = (cfg->rng_mode == ORACLE_RNG_RMT);
  for (int a = 0; a < nupd; a++) {
    const int j = upd[a];
    if (kn->kind == FMCMC_KERNEL_NMIRROR) {
      /* rnorm(k, mean = 2 mu[which.] - theta0[which.], sd = scale[which.]) :113-117 */
      const double mean = 2.0 * ks->mmu[j] - theta0[j], sd = ks->msc[j];
      const double z = (rmt && sd == 0.0) ? 0.0 : draw_normal(cfg, step, chain, (uint32_t)a);
      theta1[j] = mean + sd * z;
    } else {
      /* runif(k, min = 2 mu - theta1[which.] - sqrt3 scale, max = ... + sqrt3 scale) :246-250: mu and scale are NOT
       * subset, so the a-th updated parameter reads mu[a], scale[a] (recycling rules of runif) */
      const double sqrt3 = (cfg->math_mode == ORACLE_MATH_R) ? sqrt(3.0) : fmh_sqrt(3.0);
      const double c = 2.0 * ks->mmu[a] - theta0[j];
      const double lo = c - sqrt3 * ks->msc[a], hi = c + sqrt3 * ks->msc[a];
      const double u = (rmt && lo == hi) ? 0.0 : draw_unif(cfg, step, chain, (uint32_t)a);
      theta1[j] = lo + (hi - lo) * u;
    }
  }
  ks->abs_iter += 1;
  for (int a = 0; a < nupd; a++) {
    const int j = upd[a];
    theta1[j] = reflect1(theta1[j], kn->lb[j], kn->ub[j], cfg->math_mode);
  }
}

/* stats::cov of N rows (src/library/stats/src/cov.c, cov_complete1, pearson): long-double column means refined by a
 * second pass, then long-double sums of cross products over (N - 1). */
static void r_cov(const double* X, int N, int k, double* out) {
  long double xm[MAXK];
  for (int a = 0; a < k; a++) {
    long double s = 0.0L;
    for (int t = 0; t < N; t++) s += X[t * k + a];
    long double m = s / N;
    s = 0.0L;
    for (int t = 0; t < N; t++) s += (X[t * k + a] - m);
    xm[a] = m + s / N;
  }
  for (int a = 0; a < k; a++)
    for (int b = 0; b <= a; b++) {
      long double s = 0.0L;
      for (int t = 0; t < N; t++) s += (X[t * k + a] - xm[a]) * (X[t * k + b] - xm[b]);
      out[a * k + b] = out[b * k + a] = (double)(s / (N - 1));
    }
}

static int propose_adapt(const ocfg* cfg, const fmcmc_kernel* kn, kstate* ks, int64_t i,
                         uint32_t step, uint32_t chain, const double* theta0, double* theta1) {
  const int kf = ks->kf;
  if (kn->until > (double)ks->abs_iter && ks->abs_iter > kn->warmup && i > 2 &&
      (i % kn->freq) == 0) {
    double x[MAXK], mean_t[MAXK], Ik[MAXK * MAXK];
    for (int a = 0; a < kf; a++)
      for (int b = 0; b < kf; b++) Ik[a * kf + b] = (a == b) ? 1.0 * kn->eps : 0.0;
    if (kn->bw > 0) {
      /* Sigma <<- Sd * (cov(env$ans[(i - bw + 1):(i - 1), which.]) + Ik) :120-125 */
      const int N = kn->bw - 1;
      if (i - kn->bw + 1 < 1 || N < 2) { ks->abs_iter += 1; return FMCMC_CHAIN_BAD_WINDOW; }
      const double* X = ks->hist + (i - kn->bw) * kf; /* row i - bw + 1 */
      double cv[MAXK * MAXK];
      if (cfg->math_mode == ORACLE_MATH_R) {
        r_cov(X, N, kf, cv);
      } else { /* canonical: plain sequential column sums, then fma-accumulated centred cross products */
        double m[MAXK];
        for (int a = 0; a < kf; a++) {
          double sm = 0.0;
          for (int t = 0; t < N; t++) sm = sm + X[t * kf + a];
          m[a] = sm / (double)N;
        }
        for (int a = 0; a < kf; a++)
          for (int b = 0; b < kf; b++) {
            double acc = 0.0;
            for (int t = 0; t < N; t++) acc = fmh_fma(X[t * kf + a] - m[a], X[t * kf + b] - m[b], acc);
            cv[a * kf + b] = acc / (double)(N - 1);
          }
      }
      for (int a = 0; a < kf * kf; a++) ks->Sigma[a] = kn->Sd * (cv[a] + Ik[a]);
    } else {
      if (i - kn->freq < 1) { ks->abs_iter += 1; return FMCMC_CHAIN_BAD_WINDOW; } /* ans[0:(i-1), ] has i - 1 < freq rows: `[, , freq]` fails */
      if (!ks->have_mean) { /* colMeans(ans[1:(i-1), which.]) :130-131 */
        for (int a = 0; a < kf; a++)
          ks->mean_prev[a] = (cfg->math_mode == ORACLE_MATH_R)
                                 ? (double)(ks->run_sum_ld[a] / (long double)(i - 1))
                                 : ks->run_sum[a] / (double)(i - 1);
        ks->have_mean = 1;
      }
      /* rows (i - freq):(i - 1) folded in one by one, t. = abs_iter - freq + (row - 1) (R/recursive.R:79-108,:129-136);
       * eps = 1e-5 and Sd = 1 (the default of cov_recursive; the kernel's Sd is NOT passed) :139-156 */
      for (int j = 0; j < kn->freq; j++) {
        const double t = (double)(ks->abs_iter - kn->freq + j);
        if (kn->freq == 1) for (int a = 0; a < kf; a++) x[a] = theta0[ks->which[a]]; /* ans[i-1, which.] */
        else for (int a = 0; a < kf; a++) x[a] = ks->hist[(i - kn->freq + j - 1) * kf + a];
        fmcmc_oracle_mean_recursive(x, ks->mean_prev, t, kf, mean_t);
        fmcmc_oracle_cov_recursive(x, ks->Sigma, ks->mean_prev, mean_t, t, 1e-5, 1.0, Ik, kf);
        for (int a = 0; a < kf; a++) ks->mean_prev[a] = mean_t[a];
      }
    }
  }
  ks->abs_iter += 1;
  double z[MAXK], delta[MAXK];
  for (int a = 0; a < kf; a++) z[a] = draw_normal(cfg, step, chain, (uint32_t)a);
  if (cfg->math_mode == ORACLE_MATH_R) {
    /* MASS::mvrnorm: mu + V diag(sqrt(pmax(ev,0))) z, eigenvalues decreasing */
    double ev[MAXK], V[MAXK * MAXK];
    jacobi_eig(ks->Sigma, kf, ev, V);
    if (ev[kf - 1] < -1e-6 * fabs(ev[0])) return FMCMC_CHAIN_NOT_PD;
    for (int a = 0; a < kf; a++) {
      double s = 0.0;
      for (int b = 0; b < kf; b++) s += V[a * kf + b] * (sqrt(ev[b] > 0 ? ev[b] : 0.0) * z[b]);
      delta[a] = kn->mu[ks->which[a]] + s;
    }
  } else {
    double L[MAXK * MAXK];
    if (kn->k > 64) {   /* (more parameters than a wavefront has lanes: the one-workgroup-per-chain kernel keeps the Cholesky factor) */
      if (chol_lower_canon(ks->Sigma, L, kf)) return FMCMC_CHAIN_NOT_PD;
      for (int a = 0; a < kf; a++) {
        double s = 0.0;
        for (int b = 0; b <= a; b++) s = fmh_fma(L[a * kf + b], z[b], s);
        delta[a] = kn->mu[ks->which[a]] + s;
      }
    } else {
      double D[MAXK], u[MAXK];
      if (ldl_lower_canon(ks->Sigma, L, D, kf)) return FMCMC_CHAIN_NOT_PD;
      for (int a = 0; a < kf; a++) u[a] = fmh_sqrt(D[a]) * z[a];
      for (int a = 0; a < kf; a++) {
        double s = 0.0;
        for (int b = 0; b <= a; b++) s = fmh_fma(L[a * kf + b], u[b], s);   /* (L_aa = 1) */
        delta[a] = kn->mu[ks->which[a]] + s;
      }
    }
  }
  for (int a = 0; a < kn->k; a++) theta1[a] = theta0[a];
  for (int a = 0; a < kf; a++) theta1[ks->which[a]] = theta0[ks->which[a]] + delta[a];
  for (int a = 0; a < kf; a++) {
    int j = ks->which[a];
    theta1[j] = reflect1(theta1[j], kn->lb[j], kn->ub[j], cfg->math_mode);
  }
  return FMCMC_CHAIN_OK;
}

/* Canonical update of kernel_ram's lower factor (R/kernel_ram.R:136-146), in PRODUCT form:
 *     S (I + cp z z^T) S^T = (S T)(S T)^T,   T = chol(I + sg p p^T),  p = sqrt|cp| z,  sg = sign(cp).
 * The Cholesky factor of identity + rank one is known in closed form (Gill, Golub, Murray & Saunders 1974, "Methods
 * for modifying matrix factorizations", method C1): with beta_0 = sg, beta_{j+1} = beta_j + p_j^2,
 *     T_jj = d_j = sqrt(beta_{j+1} / beta_j),     T_ij = p_i p_j / (beta_j d_j)   (i > j),
 * hence S'_ij = S_ij d_j + G_ij kappa_j with G_ij = sum_{m = j+1..i} S_im z_m and kappa_j = |cp| z_j / (beta_j d_j).
 * Nothing sequential is left but two fma chains (the prefix sums of z^2, which also give |z|^2, and G along a row):
 * the square roots and divisions are independent per column.  |cp| |z|^2 = eta |a_n - arate| < 1, so for a downdate
 * every beta_j stays in [-1, -0.23]: the update cannot fail for finite input (the reference's chol() + nearPD path,
 * R/kernel_ram.R:141-146, is never needed); non-finite input is reported as failure and leaves S untouched.
 * Pz[j] = sum_{b<j} z_b^2 from scan_sq_canon below (Pz[k] = |z|^2), cp = eta (a_n - arate) / |z|^2.  0 ok, 1 failed. */
/* Prefix sums of z^2 in the canonical order: q_j = z_j z_j, then a Hillis-Steele scan over the index with offsets
 * 1, 2, 4, ... (every element adds the element `s` places below it, all at once) -- the order a 64-lane wavefront
 * computes in log2(k) steps.  Pz[0] = 0, Pz[j + 1] = q_0 + ... + q_j. */
static void scan_sq_canon(const double* z, int k, double* Pz) {
  double q[MAXK], t[MAXK];
  for (int j = 0; j < k; j++) q[j] = z[j] * z[j];
  for (int s = 1; s < k; s <<= 1) {
    for (int j = 0; j < k; j++) t[j] = (j >= s) ? q[j] + q[j - s] : q[j];
    for (int j = 0; j < k; j++) q[j] = t[j];
  }
  Pz[0] = 0.0;
  for (int j = 0; j < k; j++) Pz[j + 1] = q[j];
}

static int ram_factor_update_canon(double* L, const double* z, const double* Pz, double cp, int k) {
  const double acp = fmh_abs(cp), sg = (cp > 0.0) ? 1.0 : -1.0;
  double d[MAXK], kap[MAXK];
  for (int j = 0; j < k; j++) {
    const double b0 = fmh_fma(acp, Pz[j], sg), b1 = fmh_fma(acp, Pz[j + 1], sg);
    const double rho = b1 / b0;
    if (!(rho > 0.0) || !fmh_isfinite(rho)) return 1;
    d[j] = fmh_sqrt(rho);
    kap[j] = (acp * z[j]) / (b0 * d[j]);
    if (!fmh_isfinite(kap[j])) return 1;
  }
  for (int i = 0; i < k; i++) {
    double G = 0.0;
    for (int j = i; j >= 0; j--) {
      const double sij = L[i * k + j];
      L[i * k + j] = fmh_fma(G, kap[j], sij * d[j]);
      G = fmh_fma(sij, z[j], G);
    }
  }
  return 0;
}

/* kernel_ram proposal (R/kernel_ram.R:123-158). f0 = current log-posterior. */
static int propose_ram(const ocfg* cfg, const fmcmc_model* m, const fmcmc_kernel* kn, kstate* ks,
                       int64_t i, uint32_t step, uint32_t chain, const double* theta0, double f0,
                       double* theta1 /* in: previous theta1; out: new proposal */) {
  const int kf = ks->kf;
  double U[MAXK], v[MAXK];
  /* U <- qfun(k) (R/kernel_ram.R:124): the default rt(k, k) or one of the built-in families of fmcmc_kernel.ram_qfun */
  const double df = (kn->ram_qfun == FMCMC_RAM_QFUN_T_DF) ? kn->ram_df : (double)kf;
  const double eta_exp = (kn->ram_eta_exp != 0.0) ? kn->ram_eta_exp : (2.0 / 3.0);
  for (int a = 0; a < kf; a++)
    U[a] = (kn->ram_qfun == FMCMC_RAM_QFUN_NORMAL) ? draw_normal(cfg, step, chain, (uint32_t)a)
                                                   : draw_t(cfg, step, chain, (uint32_t)a, df);
  if (cfg->math_mode == ORACLE_MATH_R) {
    for (int a = 0; a < kf; a++) {
      double s = 0.0;
      for (int b = 0; b < kf; b++) s += ks->Sigma[a * kf + b] * U[b];
      v[a] = s;
    }
  } else {
    /* (S U)_a as an fma chain from the diagonal DOWN to column 0: its partial sums are the G_ab of the factor update
     * (ram_factor_update_canon), so an implementation can keep them and update S element by element */
    for (int a = 0; a < kf; a++) {
      double s = 0.0;
      for (int b = a; b >= 0; b--) s = fmh_fma(ks->Sigma[a * kf + b], U[b], s);
      v[a] = s;
    }
  }
  for (int a = 0; a < kf; a++) theta1[ks->which[a]] = theta0[ks->which[a]] + v[a];

  if (kn->until > (double)ks->abs_iter && ks->abs_iter > kn->warmup && (i % kn->freq) == 0) {
    double f1u = logpost(cfg, m, theta1); /* un-reflected proposal :132 */
    double a_n, eta;
    if (cfg->math_mode == ORACLE_MATH_R) {
      a_n = exp(f1u - f0); /* min(1, NaN) is NaN in R, then !is.finite -> 0 :132-134 */
      if (a_n > 1.0) a_n = 1.0;
      if (!isfinite(a_n)) a_n = 0.0;
      eta = pow((double)i, -eta_exp) * (double)kf; /* eta(env$i, k) :67 */
      if (eta > 1.0) eta = 1.0;
      /* Sigma %*% (Ik + eta*(a_n-arate)*UU^T/||U||^2) %*% t(Sigma); t(chol()) :136-146 */
      double nrm = 0.0;
      for (int a = 0; a < kf; a++) nrm += U[a] * U[a];
      nrm = sqrt(nrm);
      double nrm2 = nrm * nrm;
      double M[MAXK * MAXK], T[MAXK * MAXK], S2[MAXK * MAXK], L[MAXK * MAXK];
      for (int a = 0; a < kf; a++)
        for (int b = 0; b < kf; b++)
          M[a * kf + b] = ((a == b) ? 1.0 : 0.0) + eta * (a_n - kn->arate) * (U[a] * U[b]) / nrm2;
      for (int a = 0; a < kf; a++)
        for (int b = 0; b < kf; b++) {
          double s = 0.0;
          for (int c = 0; c < kf; c++) s += ks->Sigma[a * kf + c] * M[c * kf + b];
          T[a * kf + b] = s;
        }
      for (int a = 0; a < kf; a++)
        for (int b = 0; b < kf; b++) {
          double s = 0.0;
          for (int c = 0; c < kf; c++) s += T[a * kf + c] * ks->Sigma[b * kf + c];
          S2[a * kf + b] = s;
        }
      if (chol_lower_plain(S2, L, kf)) {
        ks->nerrors += 1; /* Matrix::nearPD fallback is not restated (parity unpinned) */
      } else {
        memcpy(ks->Sigma, L, sizeof(double) * kf * kf);
      }
    } else {
      a_n = fmh_exp(f1u - f0);
      if (fmh_isnan(a_n)) a_n = 0.0;
      else if (a_n > 1.0) a_n = 1.0;
      eta = (double)kf * fmh_exp((-eta_exp) * fmh_log((double)i));
      if (eta > 1.0) eta = 1.0;
      double Pz[MAXK + 1];
      scan_sq_canon(U, kf, Pz);
      double cp = (eta * (a_n - kn->arate)) / Pz[kf];
      if (cp != 0.0 && fmh_isfinite(cp)) {
        double L[MAXK * MAXK];
        memcpy(L, ks->Sigma, sizeof(double) * kf * kf);
        if (ram_factor_update_canon(L, U, Pz, cp, kf)) ks->nerrors += 1;
        else memcpy(ks->Sigma, L, sizeof(double) * kf * kf);
      }
    }
    if (kn->constr) /* Sigma <<- constr[which., which.] * Sigma :149-150 (element-wise) */
      for (int a = 0; a < kf * kf; a++) ks->Sigma[a] = kn->constr[a] * ks->Sigma[a];
  }
  ks->abs_iter += 1;
  for (int a = 0; a < kf; a++) {
    int j = ks->which[a];
    theta1[j] = reflect1(theta1[j], kn->lb[j], kn->ub[j], cfg->math_mode);
  }
  return FMCMC_CHAIN_OK;
}

/* ------------------------------------------------------------------------------------------
 * the run: MCMC_without_conv_checker over all chains of the call (R/mcmc.R:643-673 serial
 * fan-out: chains share ONE R stream sequentially) x single-chain loop (R/mcmc.R:720-838)
 * ---------------------------------------------------------------------------------------- */
int64_t fmcmc_oracle_kept_rows(int64_t nsteps, int64_t burnin, int64_t thin) {
  return (nsteps - burnin) / thin;
}

int fmcmc_oracle_run(const fmcmc_model* m, const fmcmc_kernel* kn, const fmcmc_run* run,
                     fmcmc_state* st, fmcmc_out* out, int rng_mode, int math_mode, r_rng* g) {
  const int k = kn->k;
  if (k > MAXK || k < 1) return FMCMC_ERR_ARG;
  if (run->burnin >= run->nsteps || run->thin >= run->nsteps || run->thin < 1) return FMCMC_ERR_ARG;
  if (kn->kind == FMCMC_KERNEL_ADAPT && (kn->freq < 1 || kn->bw < 0 || (kn->bw > 0 && kn->bw > kn->warmup))) return FMCMC_ERR_ARG;
  if (kn->kind == FMCMC_KERNEL_RAM && kn->freq < 1) return FMCMC_ERR_ARG;
  const int mirror = (kn->kind == FMCMC_KERNEL_NMIRROR || kn->kind == FMCMC_KERNEL_UMIRROR);
  const int simple = (kn->kind == FMCMC_KERNEL_NORMAL || kn->kind == FMCMC_KERNEL_NORMAL_REFLECTIVE ||
                      kn->kind == FMCMC_KERNEL_UNIF || kn->kind == FMCMC_KERNEL_UNIF_REFLECTIVE || mirror);
  if (simple && kn->scheme == FMCMC_SCHEME_EXPLICIT && (!kn->scheme_seq || kn->scheme_len < 1)) return FMCMC_ERR_ARG;
  ocfg cfg; cfg.rng_mode = rng_mode; cfg.math_mode = math_mode; cfg.g = g; cfg.seed = run->seed; cfg.lg_hs = NULL;
  double lg_hs[MAXK];   /* data-only sums of the canonical logistic form: once per run */
  if (m->family == FMCMC_FAM_LOGISTIC && math_mode == ORACLE_MATH_CANON) { logit_hs(m, lg_hs); cfg.lg_hs = lg_hs; }
  const int64_t C = run->nchains, nsteps = run->nsteps;
  const int64_t S = fmcmc_oracle_kept_rows(nsteps, run->burnin, run->thin);
  const int64_t nwords = (nsteps + 31) / 32;
  int any_err = 0;
  double* logu = (double*)malloc(sizeof(double) * (size_t)(nsteps + 1));
  int32_t* cols_buf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nsteps + 1));
  const int need_hist = (kn->kind == FMCMC_KERNEL_ADAPT && (kn->bw > 0 || kn->freq > 1));
  double* hist = need_hist ? (double*)malloc(sizeof(double) * (size_t)(nsteps + 1) * MAXK) : NULL;

  kstate ks;
  ks.k = k; ks.kf = 0;
  for (int a = 0; a < k; a++)
    if (!kn->fixed[a]) ks.which[ks.kf++] = a;
  const int kf = ks.kf;
  if (kf == 0) { free(logu); free(cols_buf); free(hist); return FMCMC_ERR_ARG; }
  for (int a = 0; a < k; a++)
    if (kn->kind != FMCMC_KERNEL_NORMAL && kn->kind != FMCMC_KERNEL_UNIF && !(kn->ub[a] > kn->lb[a])) {
      free(logu); free(cols_buf); free(hist); return FMCMC_ERR_ARG;
    }
  /* sample(which(!fixed), nsteps, TRUE): a length-one x >= 1 means sample(1:x) in R (R/kernel.R:110) */
  int pool[MAXK], npool = 0;
  if (kf == 1) for (int a = 0; a <= ks.which[0]; a++) pool[npool++] = a;
  else for (int a = 0; a < kf; a++) pool[npool++] = ks.which[a];

  for (int64_t c = 0; c < C; c++) {
    const uint32_t chain = (uint32_t)(run->chain_base + c);
    /* load kernel state */
    ks.abs_iter = 0; ks.have_mean = 0; ks.nerrors = 0;
    if (kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM) {
      if (st->fresh) {
        for (int a = 0; a < kf; a++)
          for (int b = 0; b < kf; b++) ks.Sigma[a * kf + b] = (a == b) ? 1.0 * kn->eps : 0.0;
      } else {
        memcpy(ks.Sigma, st->Sigma + c * kf * kf, sizeof(double) * kf * kf);
        ks.abs_iter = st->abs_iter[c];
        if (st->nerrors) ks.nerrors = st->nerrors[c];
        if (kn->kind == FMCMC_KERNEL_ADAPT) {
          ks.have_mean = st->have_mean[c];
          memcpy(ks.mean_prev, st->mean_prev + c * kf, sizeof(double) * kf);
        }
      }
    }
    double theta0[MAXK], theta1[MAXK];
    for (int a = 0; a < k; a++) theta0[a] = theta1[a] = st->theta0[c * k + a];
    if (mirror) {
      ks.nzero = 0;
      if (st->fresh) {
        for (int a = 0; a < k; a++) { ks.mmu[a] = kn->mu[a]; ks.msc[a] = kn->scale[a]; }
        for (int a = 0; a < k; a++) ks.obs_arate[a] = NAN;
      } else {
        for (int a = 0; a < k; a++) { ks.mmu[a] = st->mirror_mu[c * k + a]; ks.msc[a] = st->mirror_scale[c * k + a]; }
        for (int a = 0; a < k; a++) ks.obs_arate[a] = st->obs_arate[c * k + a];
        ks.abs_iter = st->abs_iter[c];
      }
    }

    /* R <- log(runif(nsteps)) drawn up front (R/mcmc.R:726); R[1] is never used */
    if (rng_mode == ORACLE_RNG_RMT)
      for (int64_t i = 1; i <= nsteps; i++) logu[i] = log(r_unif_rand(g));

    /* plan of scheme = "random": drawn when the kernel initialises, i.e. at the first proposal, behind R (R/kernel.R:106-113) */
    ks.cols = NULL;
    if (simple && kn->scheme == FMCMC_SCHEME_RANDOM) {
      int32_t* io = st->scheme_cols ? st->scheme_cols + c * nsteps : NULL;
      if (rng_mode == ORACLE_RNG_RMT && io && !st->fresh) {
        ks.cols = io; /* the kernel object keeps its plan across calls */
      } else {
        for (int64_t i = 1; i <= nsteps; i++)
          cols_buf[i - 1] = (rng_mode == ORACLE_RNG_RMT)
                                ? pool[(int)r_unif_index(g, (double)npool)]
                                : pool[fmh_scheme_index(cfg.seed, (uint32_t)i, chain, (uint32_t)npool)];
        if (io) memcpy(io, cols_buf, sizeof(int32_t) * (size_t)nsteps);
        ks.cols = cols_buf;
      }
    }

    double* ans = out->samples + c * k * S;
    double* drw = out->draws ? out->draws + c * k * S : NULL;
    double* lp = out->logpost ? out->logpost + c * S : NULL;
    uint32_t* bits = out->accept_bits ? out->accept_bits + c * nwords : NULL;
    if (bits) memset(bits, 0, sizeof(uint32_t) * (size_t)nwords);
    for (int64_t s = 0; s < S * k; s++) ans[s] = NAN;
    out->status[c] = FMCMC_CHAIN_OK;
    out->status_step[c] = 0;
    int64_t nacc = 0;

    double f0 = logpost(&cfg, m, theta0), f1 = f0;
    for (int a = 0; a < kf; a++) {
      ks.run_sum[a] = theta0[ks.which[a]];
      ks.run_sum_ld[a] = (long double)theta0[ks.which[a]];
    }
#define STORE_ROW(i_, th_state, th_draw, lpv)                                        \
  do {                                                                               \
    int64_t r_ = (i_);                                                               \
    if (r_ > run->burnin && ((r_ - run->burnin) % run->thin) == 0) {                 \
      int64_t s_ = (r_ - run->burnin) / run->thin - 1;                               \
      for (int a_ = 0; a_ < k; a_++) {                                               \
        ans[a_ * S + s_] = (th_state)[a_];                                           \
        if (drw) drw[a_ * S + s_] = (th_draw)[a_];                                   \
      }                                                                              \
      if (lp) lp[s_] = (lpv);                                                        \
    }                                                                                \
  } while (0)
    STORE_ROW(1, theta0, theta0, f0);
    ks.hist = hist;
    if (hist) for (int a = 0; a < kf; a++) hist[a] = theta0[ks.which[a]]; /* row 1 */

    for (int64_t i = 2; i <= nsteps; i++) {
      const uint32_t step = (uint32_t)(run->step_base + i);
      int status = FMCMC_CHAIN_OK;
      if (mirror)
        propose_mirror(&cfg, kn, &ks, i, step, chain, theta0, theta1);
      else if (simple)
        propose_normal(&cfg, kn, &ks, i, step, chain, theta0, theta1);
      else if (kn->kind == FMCMC_KERNEL_ADAPT)
        status = propose_adapt(&cfg, kn, &ks, i, step, chain, theta0, theta1);
      else
        status = propose_ram(&cfg, m, kn, &ks, i, step, chain, theta0, f0, theta1);
      if (status == FMCMC_CHAIN_OK) {
        f1 = logpost(&cfg, m, theta1);
        if (isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST; /* R/mcmc.R:758-765 */
      }
      double ratio = f1 - f0; /* kernel$logratio :768, R/kernel.R:302-303 */
      if (status == FMCMC_CHAIN_OK && isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        out->status[c] = status;
        out->status_step[c] = i;
        for (int a = 0; a < k; a++) out->status_theta[c * k + a] = theta1[a];
        any_err = 1;
        break;
      }
      double lu = (rng_mode == ORACLE_RNG_RMT) ? logu[i] : fmh_log_accept_u(cfg.seed, step, chain);
      int moved = 0;
      if (lu < ratio) { /* strict < :770 */
        if (mirror) { /* rowSums(diff(ans)^2) of the row about to be stored */
          if (math_mode == ORACLE_MATH_R) {
            long double sq = 0.0L;
            for (int a = 0; a < k; a++) sq += (long double)((theta1[a] - theta0[a]) * (theta1[a] - theta0[a]));
            moved = (sq != 0.0L);
          } else {
            double sq = 0.0;
            for (int a = 0; a < k; a++) sq = sq + (theta1[a] - theta0[a]) * (theta1[a] - theta0[a]);
            moved = (sq != 0.0);
          }
        }
        if (mirror) for (int a = 0; a < k; a++) ks.prev_row[a] = theta0[a];
        for (int a = 0; a < k; a++) theta0[a] = theta1[a];
        f0 = f1;
        nacc++;
        if (bits) bits[(i - 1) >> 5] |= (1u << ((i - 1) & 31));
      }
      else if (mirror) for (int a = 0; a < k; a++) ks.prev_row[a] = theta0[a];   /* (rejected: rows i - 1 and i are the same) */
      STORE_ROW(i, theta0, theta1, f1);
      if (mirror && !moved) ks.nzero += 1;
      if (hist) for (int a = 0; a < kf; a++) hist[(i - 1) * kf + a] = theta0[ks.which[a]]; /* row i */
      for (int a = 0; a < kf; a++) {
        ks.run_sum[a] = ks.run_sum[a] + theta0[ks.which[a]];
        ks.run_sum_ld[a] += (long double)theta0[ks.which[a]];
      }
    }
    out->accept_count[c] = nacc;
    /* write state back */
    for (int a = 0; a < k; a++) st->theta0[c * k + a] = theta0[a];
    st->f0[c] = f0;
    if (mirror) {
      for (int a = 0; a < k; a++) { st->mirror_mu[c * k + a] = ks.mmu[a]; st->mirror_scale[c * k + a] = ks.msc[a]; }
      for (int a = 0; a < k; a++) st->obs_arate[c * k + a] = ks.obs_arate[a];
      st->abs_iter[c] = ks.abs_iter;
    }
    if (kn->kind == FMCMC_KERNEL_ADAPT || kn->kind == FMCMC_KERNEL_RAM) {
      memcpy(st->Sigma + c * kf * kf, ks.Sigma, sizeof(double) * kf * kf);
      st->abs_iter[c] = ks.abs_iter;
      if (st->nerrors) st->nerrors[c] = ks.nerrors;
      if (kn->kind == FMCMC_KERNEL_ADAPT) {
        st->have_mean[c] = ks.have_mean;
        memcpy(st->mean_prev + c * kf, ks.mean_prev, sizeof(double) * kf);
      }
    }
  }
  free(logu);
  free(cols_buf);
  free(hist);
  return any_err ? FMCMC_ERR_CHAIN : FMCMC_OK;
}

/* ------------------------------------------------------------------------------------------
 * coda::gelman.diag(x, transform = FALSE, autoburnin handled by caller, multivariate = TRUE)
 * (arithmetic lives in coda, not under /root/reference; restated from Brooks & Gelman 1998 as
 * coda implements it; mpsrf pinned by README.md:315-339,:388-412)
 * x: [m][p][N] (per chain, column-major N x p).  psrf: [p] point estimates.
 * ---------------------------------------------------------------------------------------- */
int fmcmc_oracle_gelman(const double* x, int64_t mch, int p, int64_t N, double* psrf, double* mpsrf) {
  if (mch < 2 || N < 2 || p < 1 || p > MAXK) return 1;
  double* xbar = (double*)calloc((size_t)(mch * p), sizeof(double));
  double* W = (double*)calloc((size_t)(p * p), sizeof(double));
  double* B = (double*)calloc((size_t)(p * p), sizeof(double));
  double* s2 = (double*)calloc((size_t)(mch * p), sizeof(double));
  double* Sc = (double*)calloc((size_t)(p * p), sizeof(double));
  for (int64_t c = 0; c < mch; c++) {
    const double* xc = x + c * p * N;
    for (int a = 0; a < p; a++) {
      long double s = 0.0L;
      for (int64_t t = 0; t < N; t++) s += xc[a * N + t];
      xbar[c * p + a] = (double)(s / N);
    }
    for (int a = 0; a < p; a++)
      for (int b = a; b < p; b++) {
        long double s = 0.0L;
        for (int64_t t = 0; t < N; t++)
          s += (long double)(xc[a * N + t] - xbar[c * p + a]) * (xc[b * N + t] - xbar[c * p + b]);
        Sc[a * p + b] = Sc[b * p + a] = (double)(s / (N - 1));
      }
    for (int a = 0; a < p; a++) s2[c * p + a] = Sc[a * p + a];
    for (int a = 0; a < p * p; a++) W[a] += Sc[a] / (double)mch;
  }
  double* mu = (double*)calloc((size_t)p, sizeof(double));
  for (int a = 0; a < p; a++) {
    double s = 0.0;
    for (int64_t c = 0; c < mch; c++) s += xbar[c * p + a];
    mu[a] = s / (double)mch;
  }
  for (int a = 0; a < p; a++)
    for (int b = 0; b < p; b++) {
      double s = 0.0;
      for (int64_t c = 0; c < mch; c++) s += (xbar[c * p + a] - mu[a]) * (xbar[c * p + b] - mu[b]);
      B[a * p + b] = (double)N * s / (double)(mch - 1);
    }
  /* univariate psrf */
  for (int a = 0; a < p; a++) {
    double w = W[a * p + a], b = B[a * p + a];
    double ms2 = 0, mx = 0, mx2 = 0;
    for (int64_t c = 0; c < mch; c++) {
      ms2 += s2[c * p + a]; mx += xbar[c * p + a]; mx2 += xbar[c * p + a] * xbar[c * p + a];
    }
    ms2 /= mch; mx /= mch; mx2 /= mch;
    double var_s2 = 0, cov_s2_x2 = 0, cov_s2_x = 0;
    for (int64_t c = 0; c < mch; c++) {
      double d = s2[c * p + a] - ms2;
      var_s2 += d * d;
      cov_s2_x2 += d * (xbar[c * p + a] * xbar[c * p + a] - mx2);
      cov_s2_x += d * (xbar[c * p + a] - mx);
    }
    var_s2 /= (mch - 1); cov_s2_x2 /= (mch - 1); cov_s2_x /= (mch - 1);
    double var_w = var_s2 / mch;
    double var_b = (2 * b * b) / (mch - 1);
    double cov_wb = ((double)N / mch) * (cov_s2_x2 - 2 * mu[a] * cov_s2_x);
    double V = (N - 1) * w / N + (1 + 1.0 / mch) * b / N;
    double var_V = ((double)(N - 1) * (N - 1) * var_w + (1 + 1.0 / mch) * (1 + 1.0 / mch) * var_b +
                    2.0 * (N - 1) * (1 + 1.0 / mch) * cov_wb) / ((double)N * N);
    double df_V = (2 * V * V) / var_V;
    double df_adj = (df_V + 3) / (df_V + 1);
    double R2_fixed = (double)(N - 1) / N;
    double R2_random = (1 + 1.0 / mch) * (1.0 / N) * (b / w);
    psrf[a] = sqrt(df_adj * (R2_fixed + R2_random));
  }
  /* multivariate: largest eigenvalue of W^{-1} B via CW = chol(W) (upper), as coda does */
  int rc = 0;
  if (p > 1) {
    double L[MAXK * MAXK], Y[MAXK * MAXK], Z[MAXK * MAXK];
    if (chol_lower_plain(W, L, p)) {
      rc = 2;
      *mpsrf = NAN;
    } else {
      /* Y = L^{-1} B ; Z = L^{-1} Y^T  => Z = L^{-1} B L^{-T} */
      for (int col = 0; col < p; col++)
        for (int a = 0; a < p; a++) {
          double s = B[a * p + col];
          for (int b = 0; b < a; b++) s -= L[a * p + b] * Y[b * p + col];
          Y[a * p + col] = s / L[a * p + a];
        }
      for (int col = 0; col < p; col++)
        for (int a = 0; a < p; a++) {
          double s = Y[col * p + a]; /* Y^T[a][col] */
          for (int b = 0; b < a; b++) s -= L[a * p + b] * Z[b * p + col];
          Z[a * p + col] = s / L[a * p + a];
        }
      for (int a = 0; a < p; a++)
        for (int b = a + 1; b < p; b++) {
          double t = 0.5 * (Z[a * p + b] + Z[b * p + a]);
          Z[a * p + b] = Z[b * p + a] = t;
        }
      double emax = fmcmc_oracle_top_eig_sym(Z, p);
      *mpsrf = sqrt((1.0 - 1.0 / N) + (1.0 + 1.0 / p) * emax / N);
    }
  } else {
    *mpsrf = NAN;
  }
  free(xbar); free(W); free(B); free(s2); free(Sc); free(mu);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * test hooks for include/fmh_detmath.h and include/fmh_philox.h (vectorised over n inputs)
 * ---------------------------------------------------------------------------------------- */
void fmcmc_oracle_detmath(int which, const double* x, double* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    switch (which) {
      case 0: out[i] = fmh_log(x[i]); break;
      case 1: out[i] = fmh_exp(x[i]); break;
      case 2: out[i] = fmh_log1p(x[i]); break;
      case 3: out[i] = fmh_qnorm(x[i]); break;
      case 9: out[i] = fmh_log1p(fmh_exp(x[i])); break;       /* the composition (accuracy yardstick of the fused routine) */
      case 11: out[i] = fmh_logit_g(x[i]); break;    /* the canonical logistic term g(|x|), host build */
      case 12: out[i] = fmh_tan_0_halfpi(x[i]); break;
      default: out[i] = NAN;
    }
  }
}
void fmcmc_oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                         uint32_t k1, uint32_t* out4) {
  fmh_u32x4 r = fmh_philox4x32_10(c0, c1, c2, c3, k0, k1);
  for (int a = 0; a < 4; a++) out4[a] = r.v[a];
}
/* twin of the device diagnostic fmcmc_detmath_dev (which 4..8) */
void fmcmc_oracle_detmath_rng(int which, const double* x, double* out, int64_t n, uint64_t seed) {
  for (int64_t i = 0; i < n; i++) {
    uint32_t step = (uint32_t)(i & 0xffff), chain = (uint32_t)(i >> 16), j = (uint32_t)(i % 7);
    switch (which) {
      case 4: out[i] = fmh_log_accept_u(seed, step, chain); break;
      case 5: out[i] = fmh_normal(seed, step, chain, j); break;
      case 6: out[i] = fmh_student_t(seed, step, chain, j, x[i]); break;
      case 7: out[i] = fmh_sqrt(x[i]); break;
      case 8: out[i] = 1.0 / x[i]; break;
      case 10: out[i] = fmh_unif(seed, step, chain, j); break;
      default: out[i] = NAN;
    }
  }
}
/* canonical draws: kind 0 = log accept u, 1 = normal j, 2 = student t (df) j */
double fmcmc_oracle_canon_draw(int kind, uint64_t seed, uint32_t step, uint32_t chain, uint32_t j,
                               double df) {
  if (kind == 0) return fmh_log_accept_u(seed, step, chain);
  if (kind == 1) return fmh_normal(seed, step, chain, j);
  if (kind == 3) return fmh_unif(seed, step, chain, j);
  if (kind == 4) return (double)fmh_scheme_index(seed, step, chain, j); /* j = pool size */
  return fmh_student_t(seed, step, chain, j, df);
}
/* test hook: sample.int(n, size, replace = TRUE) - 1 on the R stream */
void fmcmc_oracle_r_sample_replace(r_rng* g, int n, int size, int32_t* out) {
  for (int i = 0; i < size; i++) out[i] = (int32_t)r_unif_index(g, (double)n);
}
int fmcmc_oracle_cpu_has_fma(void) { return __builtin_cpu_supports("fma") ? 1 : 0; }
