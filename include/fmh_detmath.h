/* fmh_detmath.h — deterministic fp64 math shared by the HIP engine and the CPU oracle.
 *
 * WHY THIS EXISTS
 *   fmcmc's MH loop compares  log(runif)  <  f(theta1) - f(theta0)   (R/mcmc.R:726, :770).
 *   One flipped comparison changes the whole trajectory of a chain, so the engine
 *   (gfx950 device code) and its checker (oracle/, gcc on the host) must produce
 *   bit-identical log/exp/log1p/qnorm values.  glibc and OCML do not agree in the
 *   last bit, so both sides evaluate THESE functions: only IEEE-754 +,-,*,/,sqrt and
 *   explicit fma(), all correctly rounded on both targets, in a fixed order.
 *   Compile every translation unit that includes this header with -ffp-contract=off.
 *
 *   This header is part of the numerical SPECIFICATION of the engine ("canonical
 *   math"), not of the oracle: the oracle's R-faithful mode (RNG = Mersenne-Twister,
 *   math = libm, sums = long double) does not use it.  tests/test_detmath.py checks
 *   these functions against libm/scipy independently.
 *
 * ALGORITHMS (published; restated here, no third-party source is included)
 *   fmh_log   : x = 2^k (1+f), sqrt(1/2) < 1+f <= sqrt(2); s = f/(2+f);
 *               log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)), R = degree-7 minimax in s^2
 *               (the classic Sun/FreeBSD libm decomposition).
 *   fmh_exp   : x = k ln2 + r, |r| <= ln2/2;  exp(r) = 1 + 2r/(R(r) - r) form with the
 *               degree-5 minimax c(r) = r - r^2 P(r^2).
 *   fmh_log1p : u = fl(1+x), c = exact rounding error of 1+x (Fast2Sum);
 *               log1p(x) = log(u) + c/u folded into the low-order sum of fmh_log's core.
 *   fmh_logit_g : g(u) = log(2 cosh(u / 2)), the per-observation term of the logistic family, off a per-row polynomial
 *               table (grid 1/64, degree 5); described at the function.
 *   fmh_qnorm : Wichura (1988) Algorithm AS 241, PPND16 — the routine behind R's qnorm()
 *               (R/kernel_normal.R:71 -> stats::rnorm -> norm_rand inversion), here with
 *               fma-Horner evaluation and fmh_log in the tails.
 */
#ifndef FMH_DETMATH_H
#define FMH_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define FMH_HD __host__ __device__ __forceinline__
#else
#define FMH_HD static inline
#endif

/* ---- bit casts ---------------------------------------------------------- */
FMH_HD uint64_t fmh_d2u(double x) {
  union { double d; uint64_t u; } v; v.d = x; return v.u;
}
FMH_HD double fmh_u2d(uint64_t u) {
  union { double d; uint64_t u; } v; v.u = u; return v.d;
}
/* FMH_K(c): a literal constant the device compiler must materialise AT THE USE (two 32-bit moves)
 * instead of hoisting it out of the caller's loop into a long-lived register pair: inside the
 * MH sweep kernel such hoisted constants get spilled and every reload is a memory round trip.
 * Value-transparent (the bits are unchanged), identity on the host. */
#if defined(__HIP_DEVICE_COMPILE__)
FMH_HD double fmh_opaque_(double c) { asm volatile("" : "+s"(c)); return c; }
#define FMH_K(c) fmh_opaque_(c)
#else
#define FMH_K(c) (c)
#endif
FMH_HD double fmh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
FMH_HD double fmh_sqrt(double a) { return __builtin_sqrt(a); }
FMH_HD double fmh_abs(double a) { return __builtin_fabs(a); }
FMH_HD int fmh_isnan(double a) { return a != a; }
FMH_HD int fmh_isfinite(double a) {
  return ((fmh_d2u(a) >> 52) & 0x7ffu) != 0x7ffu;
}
FMH_HD double fmh_inf(void) { return fmh_u2d(0x7ff0000000000000ull); }
FMH_HD double fmh_nan(void) { return fmh_u2d(0x7ff8000000000000ull); }

#define FMH_LN2_HI 6.93147180369123816490e-01 /* 0x3fe62e42fee00000 */
#define FMH_LN2_LO 1.90821492927058770002e-10 /* 0x3dea39ef35793c76 */
#define FMH_INV_LN2 1.44269504088896338700e+00
#define FMH_LN_SQRT_2PI 0.918938533204672741780329736406 /* log(sqrt(2*pi)) */

/* Core of log: given u = 2^k * m exactly (m in [sqrt(1/2), sqrt(2)), f = m-1) and a small
 * additive correction `extra` (0 for log, c/u for log1p), return log(u) + extra. */
/* (three pieces, so that a kernel can schedule them between other work -- mh_lat.hpp; composed, exactly the function below) */
FMH_HD double fmh_log_s_(double f) { return f / (2.0 + f); }
FMH_HD double fmh_log_R_(double s) {
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double z = s * s;
  double w = z * z;
  double t1 = w * fmh_fma(w, fmh_fma(w, FMH_K(Lg6), FMH_K(Lg4)), FMH_K(Lg2));
  double t2 = z * fmh_fma(w, fmh_fma(w, fmh_fma(w, FMH_K(Lg7), FMH_K(Lg5)), FMH_K(Lg3)), FMH_K(Lg1));
  return t2 + t1;
}
FMH_HD double fmh_log_fin_(double f, int k, double extra, double s, double R) {
  double hfsq = 0.5 * f * f;
  double dk = (double)k;
  /* log = k*ln2_hi - ((hfsq - (s*(hfsq+R) + (k*ln2_lo + extra))) - f) */
  double lo = fmh_fma(dk, FMH_K(FMH_LN2_LO), extra);
  double inner = fmh_fma(s, hfsq + R, lo);
  return dk * FMH_K(FMH_LN2_HI) - ((hfsq - inner) - f);
}
FMH_HD double fmh_log_core_(double f, int k, double extra) {
  const double s = fmh_log_s_(f);
  const double R = fmh_log_R_(s);
  return fmh_log_fin_(f, k, extra, s, R);
}

/* Split a positive finite normal/subnormal x into k and m = x / 2^k with
 * m in [sqrt(1/2), sqrt(2)); returns f = m - 1 (exact). */
FMH_HD double fmh_log_split_(double x, int* kout) {
  uint64_t ux = fmh_d2u(x);
  int k = 0;
  if ((ux >> 52) == 0) { /* subnormal: scale by 2^54 */
    x = x * 18014398509481984.0;
    ux = fmh_d2u(x);
    k = -54;
  }
  uint32_t hx = (uint32_t)(ux >> 32);
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  /* mantissa >= sqrt(2) (high word >= 0x6a09f) -> halve it */
  uint32_t i = (hx + 0x95f64u) & 0x100000u;
  uint64_t um = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
  k += (int)(i >> 20);
  *kout = k;
  return fmh_u2d(um) - 1.0;
}

FMH_HD double fmh_log(double x) {
  if (fmh_isnan(x)) return x;
  if (x < 0.0) return fmh_nan();
  if (x == 0.0) return -fmh_inf();
  if (!fmh_isfinite(x)) return x; /* +inf */
  int k;
  double f = fmh_log_split_(x, &k);
  return fmh_log_core_(f, k, 0.0);
}

/* fmh_log for a POSITIVE, FINITE, NORMAL argument (the caller checks): same bits as fmh_log(x), but
 * straight-line code, so a scheduler can overlap it with independent work. */
/* its range reduction: x = 2^k m, m in [sqrt(1/2), sqrt(2)); returns f = m - 1 (exact) */
FMH_HD double fmh_log_split_pn_(double x, int* kout) {
  uint64_t ux = fmh_d2u(x);
  uint32_t hx = (uint32_t)(ux >> 32);
  int k = (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  uint32_t i = (hx + 0x95f64u) & 0x100000u;
  uint64_t um = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
  k += (int)(i >> 20);
  *kout = k;
  return fmh_u2d(um) - 1.0;
}
FMH_HD double fmh_log_pn(double x) {
  int k;
  const double f = fmh_log_split_pn_(x, &k);
  return fmh_log_core_(f, k, 0.0);
}

FMH_HD double fmh_log1p(double x) {
  if (fmh_isnan(x)) return x;
  if (x < -1.0) return fmh_nan();
  if (x == -1.0) return -fmh_inf();
  if (!fmh_isfinite(x)) return x; /* +inf */
  if (fmh_abs(x) < 5.551115123125783e-17) return x; /* |x| < 2^-54: log1p(x) = x */
  double u = 1.0 + x;
  /* exact rounding error of the addition (Fast2Sum, larger operand first) */
  double c = (fmh_abs(x) < 1.0) ? (x - (u - 1.0)) : (1.0 - (u - x));
  int k;
  double f = fmh_log_split_(u, &k);
  return fmh_log_core_(f, k, c / u);
}

FMH_HD double fmh_exp(double x) {
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (fmh_isnan(x)) return x;
  if (x > 709.782712893383973096) return fmh_inf();
  if (x < -745.13321910194110842) return 0.0;
  if (fmh_abs(x) < 3.7252902984619140625e-09) return 1.0 + x; /* |x| < 2^-28 */
  /* k = nearest integer to x/ln2 (round half away from zero via truncation) */
  double t0 = fmh_fma(x, FMH_INV_LN2, (x < 0.0) ? -0.5 : 0.5);
  int k = (int)t0;
  double dk = (double)k;
  double hi = fmh_fma(-dk, FMH_LN2_HI, x); /* exact: k*ln2_hi has trailing zeros */
  double lo = dk * FMH_LN2_LO;
  double r = hi - lo;
  double t = r * r;
  double c = r - t * fmh_fma(t, fmh_fma(t, fmh_fma(t, fmh_fma(t, P5, P4), P3), P2), P1);
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  /* scale by 2^k without losing subnormal results: two-step for extreme k */
  if (k > 1000) {
    y = y * fmh_u2d(0x7fe0000000000000ull); /* 2^1023 */
    k -= 1023;
  } else if (k < -1000) {
    y = y * fmh_u2d(0x0360000000000000ull); /* 2^-969 (exact, still normal) */
    k += 969;
  }
  return y * fmh_u2d((uint64_t)(1023 + k) << 52);
}

/* The per-observation term of the logistic log-likelihood (round 4; replaces the softplus tail log1p(exp(-|eta|)) of rounds
 * 1-3).  R evaluates, per observation, logp = -log1p(exp(-eta)) (y = 1) or logq = -eta - log1p(exp(-eta)) (y = 0), split at
 * eta < 0 for stability (vignettes/workflow-with-fmcmc.Rmd:35-41).  Both are
 *     (y - 1/2) eta - g(|eta|),      g(u) = log(2 cosh(u / 2)) = u / 2 + log1p(exp(-u)),
 * so the log-likelihood is a dot product of the coefficients with the data-only sums  sum_i (y_i - 1/2) x_ij  MINUS
 * sum_i g(|eta_i|): the sign of eta, y, the min(.., 0) and one of the two additions leave the observation loop.  g is even,
 * smooth (nearest singularity at distance pi) and is read off a table: on the grid u_j = j / 64, j = 0..2399, ONE degree-5
 * polynomial per row in the exact reduced argument s = 64 u - j in [0, 1) (include/fmh_logit_tab.h, generated from 60-digit
 * arithmetic by tools/gen_logit_table.py).  The argument arrives ALREADY SCALED, us = 64 |eta| -- every caller forms eta with
 * the coefficients multiplied by 64 (exact), which is the same fma chain bit for bit -- so a value costs 2 + 5 operations:
 * s = fract(us), j = trunc(us), five fmas (v_fract_f64, v_cvt_u32_f64 and three 16-byte table reads on the device), against 19
 * operations + the sign / min / add around them for the softplus form.  Worst case 1.01 ulp against a 60-digit reference
 * (rounding of c0 + the last fma; tests/test_detmath.py); the composition of two libm calls R evaluates reaches 1.5.
 * For us >= 2400 (|eta| >= 37.5) g(u) = u / 2 to 0.02 ulp; NaN propagates. */
#define FMH_LG_SCALE 64.0
#define FMH_LG_HALF_INV_SCALE 0.0078125 /* 0.5 / 64 */
FMH_HD const double* fmh_lg_tab_(void) {
#include "fmh_logit_tab.h"
  return FMH_LG_TAB_;
}
FMH_HD double fmh_logit_g_scaled(double us) { /* us = 64 |eta| >= 0 (or NaN) */
  if (!(us < (double)FMH_LG_ROWS)) return (us != us) ? us : us * FMH_LG_HALF_INV_SCALE;
  const double fl = __builtin_floor(us);
  const double s = us - fl;                                     /* exact; the device's v_fract_f64 */
  const double* T = fmh_lg_tab_() + 6 * (int)fl;
  double q = fmh_fma(s, T[5], T[4]);
  q = fmh_fma(s, q, T[3]);
  q = fmh_fma(s, q, T[2]);
  q = fmh_fma(s, q, T[1]);
  return fmh_fma(s, q, T[0]);
}
/* g(|eta|) from an unscaled eta (tests, diagnostics) */
FMH_HD double fmh_logit_g(double eta) { return fmh_logit_g_scaled(fmh_abs(eta) * FMH_LG_SCALE); }


/* tan(x) for 0 <= x <= fl(pi/2): the only use is the one-off scale adaptation of the mirror kernels,
 * scale * tan(pi/2 * obs_arate) / tan(pi/2 * arate) (R/kernel_mirror.R:121-127).  sin and cos of the argument reduced to
 * [0, pi/4] by the standard minimax kernels (Sun fdlibm k_sin.c / k_cos.c coefficients, Horner with fma), then one division;
 * above pi/4 the co-function identity tan(x) = cos(y) / sin(y), y = pi/2 - x with a two-term pi/2.  Accuracy ~1 ulp
 * (tests/test_detmath.py); device bits == host bits like every routine in this header. */
FMH_HD double fmh_sin_kernel_(double x) { /* |x| <= pi/4 */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double r = fmh_fma(z, fmh_fma(z, fmh_fma(z, fmh_fma(z, fmh_fma(z, FMH_K(S6), FMH_K(S5)), FMH_K(S4)), FMH_K(S3)), FMH_K(S2)), FMH_K(S1));
  return fmh_fma(x * z, r, x);
}
FMH_HD double fmh_cos_kernel_(double x) { /* |x| <= pi/4 */
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * fmh_fma(z, fmh_fma(z, fmh_fma(z, fmh_fma(z, fmh_fma(z, FMH_K(C6), FMH_K(C5)), FMH_K(C4)), FMH_K(C3)), FMH_K(C2)), FMH_K(C1));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + z * r);
}
FMH_HD double fmh_tan_0_halfpi(double x) {
  if (!(x >= 0.0) || x > 1.5707963267948966) return fmh_nan(); /* outside the contract (incl. NaN) */
  if (x <= 0.78539816339744828) return fmh_sin_kernel_(x) / fmh_cos_kernel_(x);
  double y = (1.57079632679489655800e+00 - x) + 6.12323399573676603587e-17; /* pi/2 = hi + lo */
  return fmh_cos_kernel_(y) / fmh_sin_kernel_(y);
}

/* Standard normal quantile, Wichura AS 241 (PPND16). p in (0,1). */
FMH_HD double fmh_qnorm(double p) {
  double q = p - 0.5;
  double r, val;
  if (fmh_abs(q) <= 0.425) {
    r = 0.180625 - q * q;
    double num = fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(
        2509.0809287301226727, r, 33430.575583588128105), r, 67265.770927008700853), r,
        45921.953931549871457), r, 13731.693765509461125), r, 1971.5909503065514427), r,
        133.14166789178437745), r, 3.387132872796366608);
    double den = fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(
        5226.495278852854561, r, 28729.085735721942674), r, 39307.89580009271061), r,
        21213.794301586595867), r, 5394.1960214247511077), r, 687.1870074920579083), r,
        42.313330701600911252), r, 1.0);
    return q * num / den;
  }
  r = (q < 0.0) ? p : (1.0 - p);
  r = fmh_sqrt(-fmh_log(r));
  if (r <= 5.0) {
    r = r - 1.6;
    double num = fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(
        7.7454501427834140764e-4, r, 0.0227238449892691845833), r, 0.24178072517745061177), r,
        1.27045825245236838258), r, 3.64784832476320460504), r, 5.7694972214606914055), r,
        4.6303378461565452959), r, 1.42343711074968357734);
    double den = fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(
        1.05075007164441684324e-9, r, 5.475938084995344946e-4), r, 0.0151986665636164571966), r,
        0.14810397642748007459), r, 0.68976733498510000455), r, 1.6763848301838038494), r,
        2.05319162663775882187), r, 1.0);
    val = num / den;
  } else {
    r = r - 5.0;
    double num = fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(
        2.01033439929228813265e-7, r, 2.71155556874348757815e-5), r, 0.0012426609473880784386), r,
        0.026532189526576123093), r, 0.29656057182850489123), r, 1.7848265399172913358), r,
        5.4637849111641143699), r, 6.6579046435011037772);
    double den = fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(fmh_fma(
        2.04426310338993978564e-15, r, 1.4215117583164458887e-7), r, 1.8463183175100546818e-5), r,
        7.868691311456132591e-4), r, 0.0148753612908506148525), r, 0.13692988092273580531), r,
        0.59983220655588793769), r, 1.0);
    val = num / den;
  }
  return (q < 0.0) ? -val : val;
}

#endif /* FMH_DETMATH_H */
