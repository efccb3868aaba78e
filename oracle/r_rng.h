/* r_rng.h — TEST INFRASTRUCTURE (oracle). See r_rng.c. */
#ifndef FMCMC_ORACLE_R_RNG_H
#define FMCMC_ORACLE_R_RNG_H
#include <stdint.h>

typedef struct r_rng r_rng;

r_rng* r_rng_new(void);
void r_rng_free(r_rng* g);
uint64_t r_rng_count(const r_rng* g);
void r_set_seed(r_rng* g, uint32_t seed);
double r_unif_rand(r_rng* g);
double r_norm_rand(r_rng* g);
double r_exp_rand(r_rng* g);
double r_rgamma(r_rng* g, double a, double scale);
double r_rchisq(r_rng* g, double df);
double r_rt(r_rng* g, double df);
double r_qnorm_std(double p);
double r_unif_index(r_rng* g, double dn); /* sample.int(dn, 1) - 1, rejection sampling */
void r_runif_vec(r_rng* g, int64_t n, double* out);
void r_rnorm_vec(r_rng* g, int64_t n, double mean, double sd, double* out);
void r_rt_vec(r_rng* g, int64_t n, double df, double* out);

#endif
