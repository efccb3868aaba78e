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
 *   fmh_log1p_exp_nonpos : the softplus tail log1p(exp(a)), a <= 0, of the logistic family as ONE division-free routine
 *               (Taylor exp on the reduced argument, table-driven logarithm with 128 entries); described at the function.
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
FMH_HD double fmh_log_core_(double f, int k, double extra) {
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * fmh_fma(w, fmh_fma(w, FMH_K(Lg6), FMH_K(Lg4)), FMH_K(Lg2));
  double t2 = z * fmh_fma(w, fmh_fma(w, fmh_fma(w, FMH_K(Lg7), FMH_K(Lg5)), FMH_K(Lg3)), FMH_K(Lg1));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  double dk = (double)k;
  /* log = k*ln2_hi - ((hfsq - (s*(hfsq+R) + (k*ln2_lo + extra))) - f) */
  double lo = fmh_fma(dk, FMH_K(FMH_LN2_LO), extra);
  double inner = fmh_fma(s, hfsq + R, lo);
  return dk * FMH_K(FMH_LN2_HI) - ((hfsq - inner) - f);
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
FMH_HD double fmh_log_pn(double x) {
  uint64_t ux = fmh_d2u(x);
  uint32_t hx = (uint32_t)(ux >> 32);
  int k = (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  uint32_t i = (hx + 0x95f64u) & 0x100000u;
  uint64_t um = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
  k += (int)(i >> 20);
  return fmh_log_core_(fmh_u2d(um) - 1.0, k, 0.0);
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

/* log1p(exp(a)) for a <= 0 -- the softplus tail of the logistic log-likelihood (R: log1p(exp(x)), vignettes/
 * workflow-with-fmcmc.Rmd:37-38), one fused routine WITHOUT a division: the logistic model evaluates it n times per
 * log-posterior and the composition fmh_log1p(fmh_exp(a)) costs three fp64 divisions (12 instructions each on gfx950, five of
 * them quarter rate) out of ~100 instructions; this one is ~45.  (On the device both tables are staged in LDS: a lookup from
 * global memory is a 64-address gather.)
 *   e = exp(a):  a = (128 kk + j) ln2/128 + r, |r| <= ln2/256 (round-to-integer by adding 1.5 2^52; two-word ln2/128);
 *                exp(r) - 1 = r + r^2 q(r), q = Taylor terms up to r^5/120 (truncation 5e-19);  2^(j/128) from a table
 *                (hi + lo);  e = 2^kk (T_hi + (T_lo + T_hi (exp(r) - 1))), exact scaling (kk >= -1010 on the fast range).
 *   log1p(e):    u = fl(1 + e) in [1, 2), c = e - (u - 1) exactly (Fast2Sum);  i = top 7 mantissa bits of u;
 *                invc_i = fl(1 / (1 + i/128)), logc_i = -log(invc_i) as hi + lo (table, tools/gen_softplus_table.py);
 *                r = fma(u, invc_i, -1) in [0, 2^-7]:  log(u) = logc_i + log1p(r) EXACTLY for the tabulated doubles, and
 *                log1p(r) = r + r^2 p(r), p = Taylor terms up to r^8/8 (truncation 1.5e-18 r);  + c / u ~ c invc_i.
 *                All terms are non-negative: no cancellation; entry 0 is (1, 0, 0), so tiny e returns e.
 * e is carried as a double-double (the rounding error of its last sum is exact and joins c), so the only roundings that
 * count are those of the reduced argument and of the final sums: against the exact log1p(exp(a)) the result is within
 * 1.5 ulp (tests/test_detmath.py, 60-digit reference; 0.30 ulp on average) -- the composition of two faithfully rounded libm
 * calls that R evaluates reaches 1.5 ulp.  Arguments outside -700 <= a <= -2^-28 (and NaN) take the general functions. */
/* table and coefficients of fmh_log1p_exp_nonpos (shared with the chain-vectorised device twin in mh_common.hpp) */
FMH_HD const double* fmh_sp_tab_(void) {
  static const double FMH_SP_TAB[128 * 3] = {
  0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0,
  0x1.fc07f01fc07f0p-1, 0x1.fe02a6b106799p-8, -0x1.e44b7e3711e7fp-67,
  0x1.f81f81f81f820p-1, 0x1.fc0a8b0fc03c4p-7, -0x1.83092c5964281p-62,
  0x1.f44659e4a4271p-1, 0x1.7b91b07d5b126p-6, -0x1.6d80ab38e9430p-62,
  0x1.f07c1f07c1f08p-1, 0x1.f829b0e7832f8p-6, 0x1.33e3f04f1ef25p-60,
  0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd68p-5, -0x1.5bfa937f551b7p-59,
  0x1.e9131abf0b767p-1, 0x1.77458f632dcffp-5, 0x1.8d3ca87b92968p-63,
  0x1.e573ac901e574p-1, 0x1.b42dd711971b9p-5, 0x1.0a34531f67db5p-59,
  0x1.e1e1e1e1e1e1ep-1, 0x1.f0a30c01162a8p-5, 0x1.85f325c5bbacdp-59,
  0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae3p-4, 0x1.2189705cf74cap-58,
  0x1.dae6076b981dbp-1, 0x1.341d7961bd1d0p-4, -0x1.3599f227becbbp-58,
  0x1.d77b654b82c34p-1, 0x1.51b073f06183cp-4, -0x1.5b61c65e5741ap-58,
  0x1.d41d41d41d41dp-1, 0x1.6f0d28ae56b4ep-4, -0x1.20db323097324p-59,
  0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b23p-4, -0x1.294d2f5668495p-58,
  0x1.cd85689039b0bp-1, 0x1.a926d3a4ad562p-4, -0x1.d7a16eab1e2adp-59,
  0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4, 0x1.2eb0bf7c0b0d9p-59,
  0x1.c71c71c71c71cp-1, 0x1.e27076e2af2eap-4, -0x1.61578001e015ap-60,
  0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabcp-4, -0x1.5746b9981b36cp-58,
  0x1.c0e070381c0e0p-1, 0x1.0d77e7cd08e5bp-3, 0x1.9a5dc5e9030adp-57,
  0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a2p-3, -0x1.fbe7ee5c69946p-57,
  0x1.bacf914c1bad0p-1, 0x1.29552f81ff521p-3, 0x1.301771c407dc0p-57,
  0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f75p-3, 0x1.e6cb62af18a02p-62,
  0x1.b4e81b4e81b4fp-1, 0x1.44d2b6ccb7d1cp-3, 0x1.7d3d950f87e23p-59,
  0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3, -0x1.546ff8a470d3ap-57,
  0x1.af286bca1af28p-1, 0x1.5ff3070a793d6p-3, -0x1.bc60efafc6f6cp-58,
  0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21bp-3, 0x1.d551d97132e87p-57,
  0x1.a98ef606a63bep-1, 0x1.7ab890210d907p-3, -0x1.1072534a57e7dp-57,
  0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3, -0x1.9f7fdbfa08d9ap-57,
  0x1.a41a41a41a41ap-1, 0x1.9525a9cf456b6p-3, -0x1.26fb3e2b1d1dap-57,
  0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b561p-3, 0x1.24dc46c1ea664p-57,
  0x1.9ec8e951033d9p-1, 0x1.af3c94e80bff3p-3, 0x1.a3398064df33ep-57,
  0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd4p-3, 0x1.cfce744870f57p-58,
  0x1.999999999999ap-1, 0x1.c8ff7c79a9a20p-3, -0x1.4f689f8434011p-57,
  0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb94p-3, -0x1.a37794d03657dp-58,
  0x1.948b0fcd6e9e0p-1, 0x1.e27076e2af2e8p-3, -0x1.61578001e015ep-59,
  0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5935p-3, 0x1.e8637950dc20dp-57,
  0x1.8f9c18f9c18fap-1, 0x1.fb9186d5e3e29p-3, 0x1.355519b0de535p-57,
  0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2, -0x1.08ec217a5022dp-57,
  0x1.8acb90f6bf3aap-1, 0x1.0a324e27390e2p-2, 0x1.bdcfde8061c03p-56,
  0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad4p-2, 0x1.3f415699663ecp-63,
  0x1.8618618618618p-1, 0x1.1675cababa60fp-2, 0x1.ce63eab883727p-61,
  0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2, 0x1.9f1a39d500e3cp-56,
  0x1.8181818181818p-1, 0x1.22941fbcf7966p-2, -0x1.dbd7ac258a2bdp-58,
  0x1.7f405fd017f40p-1, 0x1.2895a13de86a4p-2, 0x1.7ad24c13f040fp-56,
  0x1.7d05f417d05f4p-1, 0x1.2e8e2bae11d31p-2, -0x1.1e99b72bd7bf2p-57,
  0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d56p-2, -0x1.16ea62c048cfbp-56,
  0x1.78a4c8178a4c8p-1, 0x1.3a64c556945eap-2, 0x1.cbcd735d03424p-60,
  0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2, -0x1.f79f6c1059cdbp-57,
  0x1.745d1745d1746p-1, 0x1.4618bc21c5ec2p-2, -0x1.7a42642661c62p-61,
  0x1.724287f46debcp-1, 0x1.4be5f957778a1p-2, -0x1.4b366b609027ap-58,
  0x1.702e05c0b8170p-1, 0x1.51aad872df82ep-2, -0x1.d8db0a7cc1543p-56,
  0x1.6e1f76b4337c7p-1, 0x1.5767717455a6cp-2, -0x1.fb2a49af933e8p-57,
  0x1.6c16c16c16c17p-1, 0x1.5d1bdbf5809cap-2, -0x1.7dc9c7c23801fp-56,
  0x1.6a13cd1537290p-1, 0x1.62c82f2b9c796p-2, -0x1.090a0dd59fe35p-58,
  0x1.6816816816817p-1, 0x1.686c81e9b14adp-2, 0x1.710af840538e3p-56,
  0x1.661ec6a5122f9p-1, 0x1.6e08eaa2ba1e4p-2, -0x1.bfb1b39ca3a0fp-56,
  0x1.642c8590b2164p-1, 0x1.739d7f6bbd007p-2, 0x1.ce24c53fad3f0p-58,
  0x1.623fa77016240p-1, 0x1.792a55fdd47a1p-2, 0x1.f057691fe9ed7p-56,
  0x1.6058160581606p-1, 0x1.7eaf83b82afc2p-2, -0x1.698b43096b576p-59,
  0x1.5e75bb8d015e7p-1, 0x1.842d1da1e8b18p-2, 0x1.54ec519784677p-56,
  0x1.5c9882b931057p-1, 0x1.89a3386c1425bp-2, 0x1.2d38c40881e0bp-57,
  0x1.5ac056b015ac0p-1, 0x1.8f11e873662c8p-2, 0x1.f85da755a61a3p-56,
  0x1.58ed2308158edp-1, 0x1.947941c2116fbp-2, 0x1.1266e8a3e8838p-57,
  0x1.571ed3c506b3ap-1, 0x1.99d958117e08ap-2, -0x1.315b444ee1f38p-56,
  0x1.5555555555555p-1, 0x1.9f323ecbf984dp-2, -0x1.a92e513217f58p-59,
  0x1.5390948f40febp-1, 0x1.a484090e5bb09p-2, 0x1.fff29adc3ad3bp-56,
  0x1.51d07eae2f815p-1, 0x1.a9cec9a9a084ap-2, -0x1.ab7b00ad0dabcp-58,
  0x1.5015015015015p-1, 0x1.af1293247786bp-2, 0x1.533844a15dc28p-58,
  0x1.4e5e0a72f0539p-1, 0x1.b44f77bcc8f64p-2, -0x1.a0892a8b38eedp-61,
  0x1.4cab88725af6ep-1, 0x1.b9858969310fdp-2, -0x1.f3827583b8877p-57,
  0x1.4afd6a052bf5bp-1, 0x1.beb4d9da71b7ap-2, 0x1.be1874deaef08p-56,
  0x1.49539e3b2d067p-1, 0x1.c3dd7a7cdad4dp-2, 0x1.7d9e0a5bd4d37p-57,
  0x1.47ae147ae147bp-1, 0x1.c8ff7c79a9a21p-2, 0x1.3097607bcbfeep-56,
  0x1.460cbc7f5cf9ap-1, 0x1.ce1af0b85f3ecp-2, -0x1.6416a1aa97b31p-57,
  0x1.446f86562d9fbp-1, 0x1.d32fe7e00ebd5p-2, 0x1.4ef6465f5f46ep-57,
  0x1.42d6625d51f87p-1, 0x1.d83e7258a2f3ep-2, 0x1.c515ba2ec9444p-58,
  0x1.4141414141414p-1, 0x1.dd46a04c1c4a1p-2, -0x1.19d95b62e2476p-62,
  0x1.3fb013fb013fbp-1, 0x1.e24881a7c6c26p-2, 0x1.05ec7a2caa523p-57,
  0x1.3e22cbce4a902p-1, 0x1.e744261d68789p-2, 0x1.cdf68dbcf2ed3p-56,
  0x1.3c995a47babe7p-1, 0x1.ec399d2468cc1p-2, -0x1.94623581958cfp-59,
  0x1.3b13b13b13b14p-1, 0x1.f128f5faf06ecp-2, -0x1.328df13bb38c2p-56,
  0x1.3991c2c187f63p-1, 0x1.f6123fa7028adp-2, 0x1.5456c3cb6cd06p-58,
  0x1.3813813813814p-1, 0x1.faf588f78f31dp-2, 0x1.cd7d9f2754362p-57,
  0x1.3698df3de0748p-1, 0x1.ffd2e0857f497p-2, -0x1.4d05f9366f27fp-59,
  0x1.3521cfb2b78c1p-1, 0x1.02552a5a5d0ffp-1, 0x1.e9c695d7ee800p-57,
  0x1.33ae45b57bcb2p-1, 0x1.04bdf9da926d2p-1, 0x1.8fe60804593bfp-56,
  0x1.323e34a2b10bfp-1, 0x1.0723e5c1cdf41p-1, -0x1.6a1a71dbba44ep-59,
  0x1.30d190130d190p-1, 0x1.0986f4f573521p-1, -0x1.37012b5805e02p-56,
  0x1.2f684bda12f68p-1, 0x1.0be72e4252a83p-1, 0x1.b4c4bdd99efffp-56,
  0x1.2e025c04b8097p-1, 0x1.0e44985d1cc8cp-1, -0x1.c546885a5a707p-59,
  0x1.2c9fb4d812ca0p-1, 0x1.109f39e2d4c96p-1, 0x1.f78fb26c2de46p-55,
  0x1.2b404ad012b40p-1, 0x1.12f719593efbdp-1, -0x1.67f6e731c1795p-56,
  0x1.29e4129e4129ep-1, 0x1.154c3d2f4d5eap-1, 0x1.98f33a3965e29p-57,
  0x1.288b01288b013p-1, 0x1.179eabbd899a0p-1, -0x1.c73e320bf059fp-58,
  0x1.27350b8812735p-1, 0x1.19ee6b467c96fp-1, -0x1.fa3422887e218p-57,
  0x1.25e22708092f1p-1, 0x1.1c3b81f713c25p-1, -0x1.0b583899021d1p-56,
  0x1.2492492492492p-1, 0x1.1e85f5e7040d1p-1, -0x1.084e99683070ep-55,
  0x1.23456789abcdfp-1, 0x1.20cdcd192ab6ep-1, -0x1.aabf0bc229014p-55,
  0x1.21fb78121fb78p-1, 0x1.23130d7bebf43p-1, -0x1.748725e374d6ep-55,
  0x1.20b470c67c0d9p-1, 0x1.2555bce98f7cap-1, 0x1.9810eb6b440f4p-55,
  0x1.1f7047dc11f70p-1, 0x1.2795e1289b11bp-1, 0x1.ade0fcf6e5a1dp-55,
  0x1.1e2ef3b3fb874p-1, 0x1.29d37fec2b08bp-1, 0x1.01735b2e9733fp-55,
  0x1.1cf06ada2811dp-1, 0x1.2c0e9ed448e8cp-1, -0x1.8a158f3917586p-55,
  0x1.1bb4a4046ed29p-1, 0x1.2e47436e40268p-1, 0x1.0950861a4886bp-55,
  0x1.1a7b9611a7b96p-1, 0x1.307d7334f10bep-1, 0x1.fdac850fab36dp-56,
  0x1.19453808ca29cp-1, 0x1.32b1339121d71p-1, 0x1.d02ab5b3d916bp-56,
  0x1.1811811811812p-1, 0x1.34e289d9ce1d2p-1, 0x1.775c96c42e729p-56,
  0x1.16e0689427379p-1, 0x1.37117b54747b6p-1, -0x1.808bf6deec882p-55,
  0x1.15b1e5f75270dp-1, 0x1.393e0d3562a1ap-1, -0x1.38eef67f2483ap-55,
  0x1.1485f0e0acd3bp-1, 0x1.3b68449fffc23p-1, 0x1.c63b7b06164dap-55,
  0x1.135c81135c811p-1, 0x1.3d9026a7156fbp-1, 0x1.0084c7a15a4f5p-58,
  0x1.12358e75d3033p-1, 0x1.3fb5b84d16f43p-1, 0x1.0a74ea82e55dfp-56,
  0x1.1111111111111p-1, 0x1.41d8fe84672afp-1, -0x1.ee6d0cf42e7fap-55,
  0x1.0fef010fef011p-1, 0x1.43f9fe2f9ce67p-1, 0x1.e1c9ee6d83b86p-55,
  0x1.0ecf56be69c90p-1, 0x1.4618bc21c5ec2p-1, 0x1.e85bd9bd99e3ap-56,
  0x1.0db20a88f4696p-1, 0x1.48353d1ea88dfp-1, -0x1.40a85d133f80bp-55,
  0x1.0c9714fbcda3bp-1, 0x1.4a4f85db03ebbp-1, -0x1.d76102e1644f2p-55,
  0x1.0b7e6ec259dc8p-1, 0x1.4c679afccee39p-1, -0x1.e971322ce7900p-57,
  0x1.0a6810a6810a7p-1, 0x1.4e7d811b75bb0p-1, -0x1.5d3d9ea6e9ea8p-55,
  0x1.0953f39010954p-1, 0x1.50913cc01686bp-1, 0x1.9e59d2d85ab62p-56,
  0x1.0842108421084p-1, 0x1.52a2d265bc5abp-1, 0x1.73be4578ad97bp-56,
  0x1.073260a47f7c6p-1, 0x1.54b2467999498p-1, 0x1.f4550a2d0f60cp-55,
  0x1.0624dd2f1a9fcp-1, 0x1.56bf9d5b3f399p-1, 0x1.11c6217363fcbp-57,
  0x1.05197f7d73404p-1, 0x1.58cadb5cd7989p-1, 0x1.624bc9764c22cp-55,
  0x1.0410410410410p-1, 0x1.5ad404c359f2dp-1, 0x1.eca6aa97c08e7p-55,
  0x1.03091b51f5e1ap-1, 0x1.5cdb1dc6c1765p-1, 0x1.47b71e2eb8419p-56,
  0x1.0204081020408p-1, 0x1.5ee02a9241676p-1, -0x1.bca7da80b6f7ep-55,
  0x1.0101010101010p-1, 0x1.60e32f44788d9p-1, -0x1.58376a5f4b135p-57,
  };
  return FMH_SP_TAB;
}
/* 2^(j/128), j = 0..127, as hi + lo */
FMH_HD const double* fmh_sp_exp_tab_(void) {
  static const double FMH_SP_EXP_TAB[128 * 2] = {
  0x1.0000000000000p+0, 0x0.0p+0,
  0x1.0163da9fb3335p+0, 0x1.b61299ab8cdb7p-54,
  0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56,
  0x1.04315e86e7f85p+0, -0x1.0a31c1977c96ep-54,
  0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55,
  0x1.0706b29ddf6dep+0, -0x1.c91dfe2b13c27p-55,
  0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57,
  0x1.09e3ecac6f383p+0, 0x1.1487818316136p-54,
  0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54,
  0x1.0cc922b7247f7p+0, 0x1.01edc16e24f71p-54,
  0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59,
  0x1.0fb66affed31bp+0, -0x1.b9bedc44ebd7bp-57,
  0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54,
  0x1.12abdc06c31ccp+0, -0x1.1b514b36ca5c7p-58,
  0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54,
  0x1.15a98c8a58e51p+0, 0x1.2406ab9eeab0ap-55,
  0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55,
  0x1.18af9388c8deap+0, -0x1.11023d1970f6cp-54,
  0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55,
  0x1.1bbe084045cd4p+0, -0x1.95386352ef607p-54,
  0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54,
  0x1.1ed5022fcd91dp+0, -0x1.1df98027bb78cp-54,
  0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55,
  0x1.21f49917ddc96p+0, 0x1.2a97e9494a5eep-55,
  0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54,
  0x1.251ce4fb2a63fp+0, 0x1.ac155bef4f4a4p-55,
  0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55,
  0x1.284dfe1f56381p+0, -0x1.a4c3a8c3f0d7ep-54,
  0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55,
  0x1.2b87fd0dad990p+0, -0x1.10adcd6381aa4p-59,
  0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54,
  0x1.2ecafa93e2f56p+0, 0x1.1ca0f45d52383p-56,
  0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55,
  0x1.32170fc4cd831p+0, 0x1.a9ce78e18047cp-55,
  0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54,
  0x1.356c55f929ff1p+0, -0x1.b5cee5c4e4628p-55,
  0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54,
  0x1.38cae6d05d866p+0, -0x1.e958d3c9904bdp-54,
  0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56,
  0x1.3c32dc313a8e5p+0, -0x1.efff8375d29c3p-54,
  0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55,
  0x1.3fa4504ac801cp+0, -0x1.7d023f956f9f3p-54,
  0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58,
  0x1.431f5d950a897p+0, -0x1.1c7dde35f7999p-55,
  0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59,
  0x1.46a41ed1d0057p+0, 0x1.c944bd1648a76p-54,
  0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56,
  0x1.4a32af0d7d3dep+0, 0x1.9cb62f3d1be56p-54,
  0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56,
  0x1.4dcb299fddd0dp+0, 0x1.8ecdbbc6a7833p-54,
  0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54,
  0x1.516daa2cf6642p+0, -0x1.f768569bd93efp-55,
  0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55,
  0x1.551a4ca5d920fp+0, -0x1.d689cefede59bp-55,
  0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54,
  0x1.58d12d497c7fdp+0, 0x1.295e15b9a1de8p-55,
  0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54,
  0x1.5c9268a5946b7p+0, 0x1.c4b1b816986a2p-60,
  0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54,
  0x1.605e1b976dc09p+0, -0x1.3e2429b56de47p-54,
  0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54,
  0x1.6434634ccc320p+0, -0x1.c483c759d8933p-55,
  0x1.6623882552225p+0, -0x1.bb60987591c34p-54,
  0x1.68155d44ca973p+0, 0x1.038ae44f73e65p-57,
  0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54,
  0x1.6c012750bdabfp+0, -0x1.2895667ff0b0dp-56,
  0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57,
  0x1.6ff7df9519484p+0, -0x1.83c0f25860ef6p-55,
  0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55,
  0x1.73f9a48a58174p+0, -0x1.0a8d96c65d53cp-54,
  0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54,
  0x1.780694fde5d3fp+0, 0x1.866b80a02162dp-54,
  0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55,
  0x1.7c1ed0130c132p+0, 0x1.f124cd1164dd6p-54,
  0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56,
  0x1.80427543e1a12p+0, -0x1.27c86626d972bp-54,
  0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54,
  0x1.8471a4623c7adp+0, -0x1.8d684a341cdfbp-55,
  0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54,
  0x1.88ac7d98a6699p+0, 0x1.994c2f37cb53ap-54,
  0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54,
  0x1.8cf3216b5448cp+0, -0x1.0d55e32e9e3aap-56,
  0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55,
  0x1.9145b0b91ffc6p+0, -0x1.dd6792e582524p-54,
  0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57,
  0x1.95a44cbc8520fp+0, -0x1.64b7c96a5f039p-56,
  0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54,
  0x1.9a0f170ca07bap+0, -0x1.173bd91cee632p-54,
  0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56,
  0x1.9e86319e32323p+0, 0x1.824ca78e64c6ep-56,
  0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54,
  0x1.a309bec4a2d33p+0, 0x1.6305c7ddc36abp-54,
  0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54,
  0x1.a799e1330b358p+0, 0x1.bcb7ecac563c7p-54,
  0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54,
  0x1.ac36bbfd3f37ap+0, -0x1.f9234cae76cd0p-55,
  0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54,
  0x1.b0e07298db666p+0, -0x1.bdef54c80e425p-54,
  0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57,
  0x1.b59728de5593ap+0, -0x1.c71dfbbba6de3p-54,
  0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56,
  0x1.ba5b030a1064ap+0, -0x1.efcd30e54292ep-54,
  0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55,
  0x1.bf2c25bd71e09p+0, -0x1.efdca3f6b9c73p-54,
  0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55,
  0x1.c40ab5fffd07ap+0, 0x1.b4537e083c60ap-54,
  0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54,
  0x1.c8f6d9406e7b5p+0, 0x1.1acbc48805c44p-56,
  0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56,
  0x1.cdf0b555dc3fap+0, -0x1.dd83b53829d72p-55,
  0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54,
  0x1.d2f87080d89f2p+0, -0x1.d487b719d8578p-54,
  0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55,
  0x1.d80e316c98398p+0, -0x1.11ec18beddfe8p-54,
  0x1.da9e603db3285p+0, 0x1.c2300696db532p-54,
  0x1.dd321f301b460p+0, 0x1.2da5778f018c3p-54,
  0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54,
  0x1.e264614f5a129p+0, -0x1.7b627817a1496p-54,
  0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55,
  0x1.e7a51fbc74c83p+0, 0x1.2d522ca0c8de2p-54,
  0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54,
  0x1.ecf482d8e67f1p+0, -0x1.c93f3b411ad8cp-54,
  0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54,
  0x1.f252b376bba97p+0, 0x1.3a1a5bf0d8e43p-54,
  0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54,
  0x1.f7bfdad9cbe14p+0, -0x1.dbb12d006350ap-54,
  0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55,
  0x1.fd3c22b8f71f1p+0, 0x1.2eb74966579e7p-57,
  };
  return FMH_SP_EXP_TAB;
}
#define FMH_SP_N_INV_LN2 0x1.71547652b82fep+7 /* 128 / ln 2 */
#define FMH_SP_LN2_N_HI 0x1.62e42fee00000p-8  /* ln 2 / 128, 32 significant bits: k * this is exact */
#define FMH_SP_LN2_N_LO 0x1.a39ef35793c76p-40
#define FMH_SP_SHIFT 0x1.8p52                 /* adding it rounds to an integer and leaves that integer in the low mantissa bits */
#define FMH_SP_E2 0x1.0000000000000p-1
#define FMH_SP_E3 0x1.5555555555555p-3
#define FMH_SP_E4 0x1.5555555555555p-5
#define FMH_SP_E5 0x1.1111111111111p-7
#define FMH_SP_L2 (-0x1.0000000000000p-1)
#define FMH_SP_L3 0x1.5555555555555p-2
#define FMH_SP_L4 (-0x1.0000000000000p-2)
#define FMH_SP_L5 0x1.999999999999ap-3
#define FMH_SP_L6 (-0x1.5555555555555p-3)
#define FMH_SP_L7 0x1.2492492492492p-3
#define FMH_SP_L8 (-0x1.0000000000000p-3)
#define FMH_SP_AMAX (-3.7252902984619140625e-09) /* -2^-28 */
#define FMH_SP_AMIN (-700.0)

FMH_HD double fmh_log1p_exp_nonpos(double a) {
  const double* FMH_SP_TAB = fmh_sp_tab_();
  const double E2 = FMH_SP_E2, E3 = FMH_SP_E3, E4 = FMH_SP_E4, E5 = FMH_SP_E5;
  const double L2 = FMH_SP_L2, L3 = FMH_SP_L3, L4 = FMH_SP_L4, L5 = FMH_SP_L5, L6 = FMH_SP_L6, L7 = FMH_SP_L7, L8 = FMH_SP_L8;
  if (!(a <= FMH_SP_AMAX) || a < FMH_SP_AMIN) return fmh_log1p(fmh_exp(a));
  /* ---- e = exp(a) = 2^kk 2^(j/128) exp(r),  a = (128 kk + j) ln2/128 + r,  |r| <= ln2/256 */
  double t = fmh_fma(a, FMH_K(FMH_SP_N_INV_LN2), FMH_K(FMH_SP_SHIFT));
  double kd = t - FMH_K(FMH_SP_SHIFT);
  int32_t ki = (int32_t)(uint32_t)fmh_d2u(t);
  double r = fmh_fma(-kd, FMH_K(FMH_SP_LN2_N_HI), a);
  r = fmh_fma(-kd, FMH_K(FMH_SP_LN2_N_LO), r);
  const double* X = fmh_sp_exp_tab_() + 2 * (ki & 127);
  double q = fmh_fma(r, FMH_K(E5), FMH_K(E4));
  q = fmh_fma(r, q, FMH_K(E3));
  q = fmh_fma(r, q, FMH_K(E2));
  double pm1 = fmh_fma(r * r, q, r);              /* exp(r) - 1 */
  double w = fmh_fma(X[0], pm1, X[1]);
  double er = X[0] + w;
  double el = w - (er - X[0]);                    /* exact: the rounding error of the last sum rides along as a low word */
  double sc = fmh_u2d((uint64_t)(1023 + (ki >> 7)) << 52);
  double e = er * sc;
  /* ---- log1p(e + el sc) */
  double u = 1.0 + e;
  double c = fmh_fma(el, sc, e - (u - 1.0));
  const double* T = FMH_SP_TAB + 3 * ((uint32_t)(fmh_d2u(u) >> 45) & 127u);
  double invc = T[0];
  double rr = fmh_fma(u, invc, -1.0);
  double p = fmh_fma(rr, FMH_K(L8), FMH_K(L7));
  p = fmh_fma(rr, p, FMH_K(L6));
  p = fmh_fma(rr, p, FMH_K(L5));
  p = fmh_fma(rr, p, FMH_K(L4));
  p = fmh_fma(rr, p, FMH_K(L3));
  p = fmh_fma(rr, p, FMH_K(L2));
  double s = fmh_fma(rr * rr, p, fmh_fma(c, invc, T[2]));
  return T[1] + (rr + s);
}


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
