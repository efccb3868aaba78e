// gelman.hip — cross-chain reduction behind convergence_gelman (R/convergence.R:191-246 ->
// coda::gelman.diag; formula restated in SURVEY.md App. A-4).
//
// Device side: per-chain window mean / covariance straight from the samples the sweep kernel left
// in HBM ([C][k][S], one chain = one column-major S x k matrix), then a fixed-order sum over the
// local chains into a small "partial" vector.  Partials of different GPUs ADD, so the only
// collective of the whole engine is one all-reduce(sum) of 1 + 5p + 2p^2 doubles per check.
// Host side: fmcmc_gelman_finish turns the reduced partial into psrf / mpsrf.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <vector>

#include "../../include/fmcmc_amd.h"

namespace {

constexpr int GT = 256;  // threads per block
constexpr int TT = 32;   // rows per LDS tile

// One block per chain. work[c] = { xbar[p], Sc[p*p] } (centred at `center` for xbar).
__global__ __launch_bounds__(GT) void gelman_chain_kernel(const double* __restrict__ samples, long long S,
                                                          int k, long long row0, long long N,
                                                          const int* __restrict__ cols, int p,
                                                          const double* __restrict__ center,
                                                          double* __restrict__ work) {
  extern __shared__ double sm[];
  double* s_mean = sm;          // [p]
  double* s_tile = sm + p;      // [p][TT+1]
  const int tid = threadIdx.x;
  const long long c = blockIdx.x;
  const double* base = samples + c * (long long)k * S;
  // pass 1: means (thread a sums its column in row order; p <= 64 columns)
  if (tid < p) {
    const double* col = base + (long long)cols[tid] * S + row0;
    double s = 0.0;
    for (long long t = 0; t < N; t++) s += col[t];
    s_mean[tid] = s / (double)N;
  }
  __syncthreads();
  // pass 2: covariance, pairs (a,b) a<=b dealt to threads; tiles staged through LDS
  const int npairs = p * (p + 1) / 2;
  constexpr int MAXPP = (FMCMC_MAX_K * (FMCMC_MAX_K + 1) / 2 + GT - 1) / GT;  // pairs per thread
  double acc[MAXPP];
#pragma unroll
  for (int q = 0; q < MAXPP; q++) acc[q] = 0.0;
  for (long long t0 = 0; t0 < N; t0 += TT) {
    const int nt = (int)((N - t0 < TT) ? (N - t0) : TT);
    for (int e = tid; e < p * TT; e += GT) {
      int a = e / TT, t = e % TT;
      s_tile[a * (TT + 1) + t] = (t < nt) ? (base[(long long)cols[a] * S + row0 + t0 + t] - s_mean[a]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXPP; q++) {
      int pr = tid + q * GT;
      if (pr < npairs) {
        // unrank pair index -> (a,b), a<=b, row-major upper triangle
        int a = 0, rem = pr;
        while (rem >= p - a) { rem -= p - a; a++; }
        int b = a + rem;
        double s = acc[q];
        for (int t = 0; t < nt; t++) s = __builtin_fma(s_tile[a * (TT + 1) + t], s_tile[b * (TT + 1) + t], s);
        acc[q] = s;
      }
    }
    __syncthreads();
  }
  double* w = work + c * (long long)(p + p * p);
  if (tid < p) w[tid] = s_mean[tid] - (center ? center[tid] : 0.0);
#pragma unroll
  for (int q = 0; q < MAXPP; q++) {
    int pr = tid + q * GT;
    if (pr < npairs) {
      int a = 0, rem = pr;
      while (rem >= p - a) { rem -= p - a; a++; }
      int b = a + rem;
      double v = acc[q] / (double)(N - 1);
      w[p + a * p + b] = v;
      w[p + b * p + a] = v;
    }
  }
}

// Single block: fixed-order sum over chains -> partial.
__global__ __launch_bounds__(GT) void gelman_sum_kernel(const double* __restrict__ work, long long C, int p,
                                                        double* __restrict__ partial) {
  const int tid = threadIdx.x;
  const int stride = p + p * p;
  const int len = 1 + 5 * p + 2 * p * p;
  for (int e = tid; e < len; e += GT) {
    double s = 0.0;
    if (e == 0) {
      s = (double)C;
    } else if (e < 1 + p) {  // sum xbar
      int a = e - 1;
      for (long long c = 0; c < C; c++) s += work[c * stride + a];
    } else if (e < 1 + p + p * p) {  // sum xbar xbar^T
      int ab = e - 1 - p, a = ab / p, b = ab % p;
      for (long long c = 0; c < C; c++) s = __builtin_fma(work[c * stride + a], work[c * stride + b], s);
    } else if (e < 1 + p + 2 * p * p) {  // sum S_c
      int ab = e - 1 - p - p * p;
      for (long long c = 0; c < C; c++) s += work[c * stride + p + ab];
    } else {
      int r = e - 1 - p - 2 * p * p, which = r / p, a = r % p;
      for (long long c = 0; c < C; c++) {
        double s2 = work[c * stride + p + a * p + a], xb = work[c * stride + a];
        double v = (which == 0) ? s2 : (which == 1) ? s2 * s2 : (which == 2) ? s2 * xb : s2 * xb * xb;
        s += v;
      }
    }
    partial[e] = s;
  }
}

}  // namespace

extern "C" {

int64_t fmcmc_gelman_partial_len(int32_t p) { return 1 + 5 * (int64_t)p + 2 * (int64_t)p * p; }
int64_t fmcmc_gelman_work_len(int64_t nchains, int32_t p) { return nchains * ((int64_t)p + (int64_t)p * p); }

int fmcmc_gelman_partial_dev(const double* samples, int64_t nchains, int32_t k, int64_t S, int64_t row0,
                             int64_t N, const int32_t* cols, int32_t p, const double* center,
                             double* work, double* partial, void* hip_stream) {
  if (!samples || !cols || !work || !partial || p < 1 || p > FMCMC_MAX_K || nchains < 1 || N < 2 ||
      row0 < 0 || row0 + N > S)
    return FMCMC_ERR_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  size_t lds = sizeof(double) * ((size_t)p + (size_t)p * (TT + 1));
  hipLaunchKernelGGL(gelman_chain_kernel, dim3((unsigned)nchains), dim3(GT), lds, st, samples,
                     (long long)S, (int)k, (long long)row0, (long long)N, cols, (int)p, center, work);
  hipLaunchKernelGGL(gelman_sum_kernel, dim3(1), dim3(GT), 0, st, work, (long long)nchains, (int)p, partial);
  return hipGetLastError() == hipSuccess ? FMCMC_OK : FMCMC_ERR_DEVICE;
}

// Host finish. partial is the (all-reduced) vector; xbar sums are relative to `center`.
// psrf[p] point estimates (coda's "Point est."), *mpsrf multivariate (NaN when p == 1).
int fmcmc_gelman_finish(const double* P, int32_t p, int64_t N, double* psrf, double* mpsrf) {
  if (!P || p < 1 || p > FMCMC_MAX_K || N < 2) return FMCMC_ERR_ARG;
  const double m = P[0];
  if (m < 2) return FMCMC_ERR_ARG;
  const double* sx = P + 1;
  const double* sxx = sx + p;
  const double* sS = sxx + p * p;
  const double* s_s2 = sS + p * p;
  const double* s_s2s2 = s_s2 + p;
  const double* s_s2x = s_s2s2 + p;
  const double* s_s2xx = s_s2x + p;
  std::vector<double> W(p * p), B(p * p), mu(p);
  for (int a = 0; a < p; a++) mu[a] = sx[a] / m;
  for (int a = 0; a < p; a++)
    for (int b = 0; b < p; b++) {
      W[a * p + b] = sS[a * p + b] / m;
      B[a * p + b] = (double)N * (sxx[a * p + b] - m * mu[a] * mu[b]) / (m - 1);
    }
  for (int a = 0; a < p; a++) {
    double w = W[a * p + a], b = B[a * p + a];
    double ms2 = s_s2[a] / m;
    double var_s2 = (s_s2s2[a] - m * ms2 * ms2) / (m - 1);
    double mx2 = sxx[a * p + a] / m;
    double cov_s2_x2 = (s_s2xx[a] - m * ms2 * mx2) / (m - 1);
    double cov_s2_x = (s_s2x[a] - m * ms2 * mu[a]) / (m - 1);
    double var_w = var_s2 / m;
    double var_b = (2 * b * b) / (m - 1);
    double cov_wb = ((double)N / m) * (cov_s2_x2 - 2 * mu[a] * cov_s2_x);
    double V = (N - 1) * w / N + (1 + 1.0 / m) * b / N;
    double var_V = ((double)(N - 1) * (N - 1) * var_w + (1 + 1.0 / m) * (1 + 1.0 / m) * var_b +
                    2.0 * (N - 1) * (1 + 1.0 / m) * cov_wb) / ((double)N * N);
    double df_V = (2 * V * V) / var_V;
    double df_adj = (df_V + 3) / (df_V + 1);
    psrf[a] = sqrt(df_adj * ((double)(N - 1) / N + (1 + 1.0 / m) * (1.0 / N) * (b / w)));
  }
  *mpsrf = NAN;
  if (p > 1) {
    // L = chol(W) lower; Z = L^-1 B L^-T; largest eigenvalue by cyclic Jacobi
    std::vector<double> L(p * p, 0.0), Y(p * p), Z(p * p);
    for (int j = 0; j < p; j++) {
      double d = W[j * p + j];
      for (int b = 0; b < j; b++) d -= L[j * p + b] * L[j * p + b];
      if (!(d > 0.0)) return FMCMC_ERR_CHAIN;  // gelman.diag fails -> "not converged" (R/convergence.R:207-217)
      L[j * p + j] = sqrt(d);
      for (int i = j + 1; i < p; i++) {
        double s = W[i * p + j];
        for (int b = 0; b < j; b++) s -= L[i * p + b] * L[j * p + b];
        L[i * p + j] = s / L[j * p + j];
      }
    }
    for (int col = 0; col < p; col++)
      for (int a = 0; a < p; a++) {
        double s = B[a * p + col];
        for (int b = 0; b < a; b++) s -= L[a * p + b] * Y[b * p + col];
        Y[a * p + col] = s / L[a * p + a];
      }
    for (int col = 0; col < p; col++)
      for (int a = 0; a < p; a++) {
        double s = Y[col * p + a];
        for (int b = 0; b < a; b++) s -= L[a * p + b] * Z[b * p + col];
        Z[a * p + col] = s / L[a * p + a];
      }
    for (int a = 0; a < p; a++)
      for (int b = a + 1; b < p; b++) Z[a * p + b] = Z[b * p + a] = 0.5 * (Z[a * p + b] + Z[b * p + a]);
    for (int sweep = 0; sweep < 100; sweep++) {
      double off = 0.0;
      for (int a = 0; a < p; a++)
        for (int b = a + 1; b < p; b++) off += Z[a * p + b] * Z[a * p + b];
      if (off < 1e-300) break;
      for (int i = 0; i < p; i++)
        for (int j = i + 1; j < p; j++) {
          double apq = Z[i * p + j];
          if (fabs(apq) < 1e-300) continue;
          double th = (Z[j * p + j] - Z[i * p + i]) / (2.0 * apq);
          double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
          double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
          for (int r = 0; r < p; r++) {
            double x = Z[r * p + i], y = Z[r * p + j];
            Z[r * p + i] = c * x - s * y;
            Z[r * p + j] = s * x + c * y;
          }
          for (int r = 0; r < p; r++) {
            double x = Z[i * p + r], y = Z[j * p + r];
            Z[i * p + r] = c * x - s * y;
            Z[j * p + r] = s * x + c * y;
          }
        }
    }
    double emax = Z[0];
    for (int a = 1; a < p; a++)
      if (Z[a * p + a] > emax) emax = Z[a * p + a];
    *mpsrf = sqrt((1.0 - 1.0 / N) + (1.0 + 1.0 / p) * emax / N);
  }
  return FMCMC_OK;
}

}  // extern "C"
