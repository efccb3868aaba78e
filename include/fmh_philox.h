/* fmh_philox.h — counter-based RNG of the engine's canonical ("PHILOX") stream.
 *
 * Replaces, on the device, the draws fmcmc takes from R's global generator:
 *   R/mcmc.R:726          R <- log(runif(nsteps))      -> slot 0 of every step
 *   R/kernel_normal.R:71  rnorm(k, mu, scale)          -> slots 1.. (one N(0,1) per free parameter)
 *   R/kernel_adapt.R:175  MASS::mvrnorm (k std normals) -> slots 1..
 *   R/kernel_ram.R:68     rt(k, k)                     -> slots 1.. (normal) and 64.. (chi-square)
 * R's stream is serial (Mersenne-Twister); a GPU needs random access, so the canonical
 * stream is Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11) keyed by the user seed and indexed by (chain, step, slot).  The
 * oracle implements both this stream (to be the engine's bit-exact twin) and R's own
 * generator (to be fmcmc's bit-exact twin); see oracle/README in oracle/fmcmc_oracle.c.
 *
 * Stream layout (all indices are GLOBAL, so results do not depend on how chains are
 * sharded over GPUs or how a run is cut into launches):
 *   key     = (seed_lo32, seed_hi32)
 *   counter = (step, chain, block, stream)
 *     step   : absolute 1-based MH iteration index of the chain's whole history
 *              (R's loop index i plus the number of iterations in earlier bulks)
 *     chain  : global chain id (0-based)
 *     block  : which 128-bit block of the (step,chain) substream
 *     stream : FMH_STREAM_* tag
 *   One block gives two uniforms u0,u1 in (0,1): 52 random bits each, centred,
 *   u = (bits + 0.5) * 2^-52, so log(u) and qnorm(u) are always finite.
 */
#ifndef FMH_PHILOX_H
#define FMH_PHILOX_H

#include <stdint.h>
#include "fmh_detmath.h"

#define FMH_STREAM_ACCEPT 0u   /* block 0: u0 = accept uniform */
#define FMH_STREAM_NORMAL 1u   /* block j/2, lane j%2: N(0,1) for free parameter j */
#define FMH_STREAM_GAMMA 2u    /* block = j*FMH_GAMMA_TRIES + attempt: (normal, uniform) pair */
#define FMH_STREAM_UNIF 3u     /* block j/2, lane j%2: U(0,1) for free parameter j (uniform kernels) */
#define FMH_STREAM_SCHEME 4u   /* block 0, word 0: update column of plan row `step` (scheme = "random") */
#define FMH_GAMMA_TRIES 64u

typedef struct { uint32_t v[4]; } fmh_u32x4;

FMH_HD uint32_t fmh_mulhi32_(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

FMH_HD fmh_u32x4 fmh_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = fmh_mulhi32_(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = fmh_mulhi32_(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  fmh_u32x4 out;
  out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
  return out;
}

/* 52-bit centred uniform in (0,1) from two 32-bit words (hi word first). */
FMH_HD double fmh_u01(uint32_t hi, uint32_t lo) {
  uint64_t bits = (((uint64_t)hi << 32) | (uint64_t)lo) >> 12;
  return ((double)bits + 0.5) * 2.220446049250313080847263336181640625e-16; /* 2^-52 */
}

/* The two uniforms of block (step, chain, block, stream) under seed. */
FMH_HD void fmh_uniform2(uint64_t seed, uint32_t step, uint32_t chain, uint32_t block,
                         uint32_t stream, double* u0, double* u1) {
  fmh_u32x4 r = fmh_philox4x32_10(step, chain, block, stream, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
  *u0 = fmh_u01(r.v[0], r.v[1]);
  *u1 = fmh_u01(r.v[2], r.v[3]);
}

/* log of the accept uniform of (step, chain): the canonical R[i] of R/mcmc.R:726. */
FMH_HD double fmh_log_accept_u(uint64_t seed, uint32_t step, uint32_t chain) {
  double u0, u1;
  fmh_uniform2(seed, step, chain, 0u, FMH_STREAM_ACCEPT, &u0, &u1);
  (void)u1;
  return fmh_log(u0);
}

/* j-th standard normal of (step, chain). */
FMH_HD double fmh_normal(uint64_t seed, uint32_t step, uint32_t chain, uint32_t j) {
  double u0, u1;
  fmh_uniform2(seed, step, chain, j >> 1, FMH_STREAM_NORMAL, &u0, &u1);
  return fmh_qnorm((j & 1u) ? u1 : u0);
}

/* j-th U(0,1) proposal variate of (step, chain): the canonical unif_rand() behind runif (R/kernel_unif.R:74). */
FMH_HD double fmh_unif(uint64_t seed, uint32_t step, uint32_t chain, uint32_t j) {
  double u0, u1;
  fmh_uniform2(seed, step, chain, j >> 1, FMH_STREAM_UNIF, &u0, &u1);
  return (j & 1u) ? u1 : u0;
}

/* Index in [0, npool) of plan row `row` (the LOCAL loop index i: the reference builds the plan once per kernel object
 * and reuses its rows in every later call, R/kernel.R:106-113) of chain `chain`: multiply-shift of one 32-bit word. */
FMH_HD uint32_t fmh_scheme_index(uint64_t seed, uint32_t row, uint32_t chain, uint32_t npool) {
  fmh_u32x4 r = fmh_philox4x32_10(row, chain, 0u, FMH_STREAM_SCHEME, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (uint32_t)(((uint64_t)r.v[0] * (uint64_t)npool) >> 32);
}

/* j-th chi-square(df) variate of (step, chain), df >= 1 (not necessarily integer).
 * Marsaglia & Tsang (2000) "A simple method for generating gamma variables":
 * gamma(a), a>=1: d=a-1/3, c=1/sqrt(9d); x~N(0,1), v=(1+cx)^3, accept if
 * log(u) < x^2/2 + d - d v + d log v.  a<1 uses gamma(a+1) * u^(1/a).
 * Attempt t uses block j*FMH_GAMMA_TRIES+t (u0 -> normal, u1 -> uniform); the loop is
 * bounded (acceptance > 95% per attempt) and falls back to the mean d*1 on exhaustion. */
FMH_HD double fmh_chisq(uint64_t seed, uint32_t step, uint32_t chain, uint32_t j, double df) {
  double a = 0.5 * df;
  double boost = 1.0;
  if (a < 1.0) {
    double u0, u1;
    fmh_uniform2(seed, step, chain, j * FMH_GAMMA_TRIES + (FMH_GAMMA_TRIES - 1u),
                 FMH_STREAM_GAMMA, &u0, &u1);
    (void)u1;
    boost = fmh_exp(fmh_log(u0) / a);
    a = a + 1.0;
  }
  double d = a - 1.0 / 3.0;
  double c = 1.0 / fmh_sqrt(9.0 * d);
  double g = d;
  for (uint32_t t = 0; t + 1u < FMH_GAMMA_TRIES; ++t) {
    double u0, u1;
    fmh_uniform2(seed, step, chain, j * FMH_GAMMA_TRIES + t, FMH_STREAM_GAMMA, &u0, &u1);
    double x = fmh_qnorm(u0);
    double v = fmh_fma(c, x, 1.0);
    if (v <= 0.0) continue;
    v = v * v * v;
    double lhs = fmh_log(u1);
    double rhs = fmh_fma(0.5 * x, x, d) - d * v + d * fmh_log(v);
    if (lhs < rhs) { g = d * v; break; }
  }
  return 2.0 * g * boost;
}

/* j-th Student-t(df) variate of (step, chain): N(0,1) / sqrt(chisq(df)/df), the
 * construction of R's rt() (R/kernel_ram.R:68). */
FMH_HD double fmh_student_t(uint64_t seed, uint32_t step, uint32_t chain, uint32_t j, double df) {
  double z = fmh_normal(seed, step, chain, j);
  double x2 = fmh_chisq(seed, step, chain, j, df);
  return z / fmh_sqrt(x2 / df);
}

#endif /* FMH_PHILOX_H */
