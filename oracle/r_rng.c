/* r_rng.c — TEST INFRASTRUCTURE (oracle).  Restatement of base R's default random number
 * generation, which fmcmc's hot path draws from but which is NOT under /root/reference:
 *
 *   R/mcmc.R:456   set.seed(seed)
 *   R/mcmc.R:726   log(stats::runif(nsteps))
 *   R/kernel_normal.R:71,156   stats::rnorm(k, mean, sd)
 *   R/kernel_ram.R:68          stats::rt(k, k)
 *   R/kernel_adapt.R:175       MASS::mvrnorm -> rnorm(p)
 *
 * Dependency: base R (stats/nmath), version unpinned by the reference (DESCRIPTION:36-42);
 * RNGkind defaults since R 3.6.0: "Mersenne-Twister", "Inversion", "Rejection".
 * Published algorithms restated here:
 *   - MT19937: Matsumoto & Nishimura (1998), with R's seeding scramble
 *     (LCG 69069*s+1, 50 warm-up steps, then 625 words; position forced to 624).
 *   - unif_rand(): 32-bit output * 2^-32 (2.3283064365386963e-10) clamped away from 0 and 1.
 *   - norm_rand(), INVERSION: u = floor(2^27 u1) + u2; qnorm(u / 2^27); qnorm = Wichura AS241.
 *   - exp_rand(): Ahrens & Dieter (1972) algorithm SA.
 *   - rgamma(a>=1): Ahrens & Dieter (1982) algorithm GD; a<1: Ahrens & Dieter (1974) GS.
 *   - rchisq(df) = rgamma(df/2, 2); rt(df) = norm_rand()/sqrt(rchisq(df)/df).
 * Pinned by: KATs in tests/test_oracle_rrng.py (set.seed(1); runif(3) etc.)
 * and, end to end, by the reference's printed outputs G1-G5 (tests/test_oracle_golden.py).
 */
#include "r_rng.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

struct r_rng {
  uint32_t mt[MT_N];
  int mti;
  uint64_t n_unif; /* number of unif_rand() calls so far (for draw-order tests) */
};

r_rng* r_rng_new(void) {
  r_rng* g = (r_rng*)calloc(1, sizeof(r_rng));
  if (g) r_set_seed(g, 0);
  return g;
}
void r_rng_free(r_rng* g) { free(g); }
uint64_t r_rng_count(const r_rng* g) { return g->n_unif; }

void r_set_seed(r_rng* g, uint32_t seed) {
  /* Randomize(): initial scrambling */
  for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
  /* RNG_Init(): i_seed[0..624]; FixupSeeds: i_seed[0] := 624 (mti = N) */
  seed = 69069u * seed + 1u; /* i_seed[0], overwritten by the position */
  for (int j = 0; j < MT_N; j++) {
    seed = 69069u * seed + 1u;
    g->mt[j] = seed;
  }
  g->mti = MT_N;
  g->n_unif = 0;
}

static uint32_t mt_next(r_rng* g) {
  static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
  uint32_t y;
  uint32_t* mt = g->mt;
  if (g->mti >= MT_N) {
    int kk;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 0x1u];
    }
    for (; kk < MT_N - 1; kk++) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 0x1u];
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 0x1u];
    g->mti = 0;
  }
  y = mt[g->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

double r_unif_rand(r_rng* g) {
  const double i2_32m1 = 2.328306437080797e-10; /* 1/(2^32 - 1) */
  double x = (double)mt_next(g) * 2.3283064365386963e-10; /* in [0,1) */
  g->n_unif++;
  if (x <= 0.0) return 0.5 * i2_32m1;
  if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * i2_32m1;
  return x;
}

/* Wichura AS241 PPND16 as R evaluates it: plain Horner (no fma), libm log/sqrt. */
double r_qnorm_std(double p) {
  double q = p - 0.5, r, val;
  if (fabs(q) <= 0.425) {
    r = .180625 - q * q;
    val = q *
          (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r +
               45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
            133.14166789178437745) * r + 3.387132872796366608) /
          (((((((r * 5226.495278852854561 + 28729.085735721942674) * r + 39307.89580009271061) * r +
               21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
            42.313330701600911252) * r + 1.);
    return val;
  }
  r = (q < 0) ? p : 1.0 - p;
  r = sqrt(-log(r));
  if (r <= 5.) {
    r += -1.6;
    val = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r +
               1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
            4.6303378461565452959) * r + 1.42343711074968357734) /
          (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r +
               .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r +
            2.05319162663775882187) * r + 1.);
  } else {
    r += -5.;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + .0012426609473880784386) * r +
               .026532189526576123093) * r + .29656057182850489123) * r + 1.7848265399172913358) * r +
            5.4637849111641143699) * r + 6.6579046435011037772) /
          (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
               7.868691311456132591e-4) * r + .0148753612908506148525) * r + .13692988092273580531) * r +
            .59983220655588793769) * r + 1.);
  }
  if (q < 0.0) val = -val;
  return val;
}

double r_norm_rand(r_rng* g) {
  const double BIG = 134217728.0; /* 2^27 */
  double u = r_unif_rand(g);
  u = (double)(int)(BIG * u) + r_unif_rand(g);
  return r_qnorm_std(u / BIG);
}

double r_exp_rand(r_rng* g) {
  /* q[k-1] = sum_{i=1..k} (ln 2)^i / i! */
  static const double q[] = {0.6931471805599453, 0.9333736875190459, 0.9888777961838675,
                             0.9984589039328340, 0.9998292811061389, 0.9999833164100727,
                             0.9999985691438767, 0.9999998906925558, 0.9999999924734159,
                             0.9999999995283275, 0.9999999999728814, 0.9999999999985598,
                             0.9999999999999289, 0.9999999999999968, 0.9999999999999999,
                             1.0000000000000000};
  double a = 0.;
  double u = r_unif_rand(g);
  while (u <= 0. || u >= 1.) u = r_unif_rand(g);
  for (;;) {
    u += u;
    if (u > 1.) break;
    a += q[0];
  }
  u -= 1.;
  if (u <= q[0]) return a + u;
  int i = 0;
  double ustar = r_unif_rand(g), umin = ustar;
  do {
    ustar = r_unif_rand(g);
    if (umin > ustar) umin = ustar;
    i++;
  } while (u > q[i]);
  return a + umin * q[0];
}

double r_rgamma(r_rng* g, double a, double scale) {
  const double sqrt32 = 5.656854;
  const double exp_m1 = 0.36787944117144233;
  const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191, q4 = 0.00144121,
               q5 = -7.388e-5, q6 = 2.4511e-4, q7 = 2.424e-4;
  const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062, a4 = -0.1662921,
               a5 = 0.1423657, a6 = -0.1367177, a7 = 0.1233795;
  double e, p, q, r, t, u, v, w, x, ret_val;
  double s, s2, d, q0, b, si, c;

  if (isnan(a) || isnan(scale) || a <= 0.0 || scale <= 0.0) {
    if (scale == 0. || a == 0.) return 0.;
    return NAN;
  }
  if (a < 1) { /* GS */
    e = 1.0 + exp_m1 * a;
    for (;;) {
      p = e * r_unif_rand(g);
      if (p >= 1.0) {
        x = -log((e - p) / a);
        if (r_exp_rand(g) >= (1.0 - a) * log(x)) break;
      } else {
        x = exp(log(p) / a);
        if (r_exp_rand(g) >= x) break;
      }
    }
    return scale * x;
  }
  /* GD. Step 1 */
  s2 = a - 0.5;
  s = sqrt(s2);
  d = sqrt32 - s * 12;
  /* Step 2 */
  t = r_norm_rand(g);
  x = s + 0.5 * t;
  ret_val = x * x;
  if (t >= 0) return scale * ret_val;
  /* Step 3 */
  u = r_unif_rand(g);
  if (d * u <= t * t * t) return scale * ret_val;
  /* Step 4 */
  r = 1 / a;
  q0 = ((((((q7 * r + q6) * r + q5) * r + q4) * r + q3) * r + q2) * r + q1) * r;
  if (a <= 3.686) {
    b = 0.463 + s + 0.178 * s2;
    si = 1.235;
    c = 0.195 / s - 0.079 + 0.16 * s;
  } else if (a <= 13.022) {
    b = 1.654 + 0.0076 * s2;
    si = 1.68 / s + 0.275;
    c = 0.062 / s + 0.024;
  } else {
    b = 1.77;
    si = 0.75;
    c = 0.1515 / s;
  }
  /* Step 5-7 */
  if (x > 0.0) {
    v = t / (s + s);
    if (fabs(v) <= 0.25)
      q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
    else
      q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
    if (log(1.0 - u) <= q) return scale * ret_val;
  }
  for (;;) {
    /* Step 8 */
    e = r_exp_rand(g);
    u = r_unif_rand(g);
    u = u + u - 1.0;
    if (u < 0.0)
      t = b - si * e;
    else
      t = b + si * e;
    /* Step 9 */
    if (t >= -0.71874483771719) {
      v = t / (s + s);
      if (fabs(v) <= 0.25)
        q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
      else
        q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
      if (q > 0.0) {
        w = expm1(q);
        if (c * fabs(u) <= w * exp(e - 0.5 * t * t)) break;
      }
    }
  }
  x = s + 0.5 * t;
  return scale * x * x;
}

double r_rchisq(r_rng* g, double df) { return r_rgamma(g, df / 2.0, 2.0); }

double r_rt(r_rng* g, double df) {
  if (isnan(df) || df <= 0.0) return NAN;
  if (!isfinite(df)) return r_norm_rand(g);
  double num = r_norm_rand(g);
  return num / sqrt(r_rchisq(g, df) / df);
}

/* Vector helpers for tests / data generation (R's element order). */
void r_runif_vec(r_rng* g, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = r_unif_rand(g);
}
void r_rnorm_vec(r_rng* g, int64_t n, double mean, double sd, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = mean + sd * r_norm_rand(g);
}
void r_rt_vec(r_rng* g, int64_t n, double df, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = r_rt(g, df);
}

/* R_unif_index (src/main/RNG.c, sample.kind = "Rejection", the default since R 3.6.0): uniform integer in [0, dn)
 * by rejection from the next power of two; rbits() consumes one unif_rand per 16 bits (and one even for 0 bits). */
static double r_rbits(r_rng* g, int bits) {
  int64_t v = 0;
  for (int n = 0; n <= bits; n += 16) {
    int v1 = (int)floor(r_unif_rand(g) * 65536);
    v = 65536 * v + v1;
  }
  if (bits < 64) v &= (((int64_t)1 << bits) - 1);
  return (double)v;
}
double r_unif_index(r_rng* g, double dn) {
  if (dn <= 0) return 0.0;
  int bits = (int)ceil(log2(dn));
  double dv;
  do { dv = r_rbits(g, bits); } while (dn <= dv);
  return dv;
}
