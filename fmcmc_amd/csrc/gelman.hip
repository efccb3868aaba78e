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

// One block (4 wavefronts) per chain: work[c] = { xbar[p] - center, S_c[p*p] } of the window [row0, row0 + N).
//
// The window of one chain is a column-major N x p matrix D; its covariance is the rank-N update D'D -- the one GEMM-shaped
// piece of the whole engine besides the log-posterior, and at config C4 (512 chains x p = 50 x 5,000 rows: 1 GB, 13 GFLOP)
// it is as much arithmetic as memory traffic.  So it runs on the matrix cores: v_mfma_f64_16x16x4 with M = N = a block of
// 16 columns and K = 4 consecutive rows.  Both operands have the SAME lane layout (lane l holds row t0 + l / 16 of column
// 16 cb + l % 16), so a K-step loads one double per lane and column block and feeds all NCB (NCB + 1) / 2 tile pairs.
// ONE pass over the data: rows are shifted by the chain's first window row (d = x - x[row0]; the shift is within a few
// standard deviations of the mean, which keeps sum d d' - N dbar dbar' free of cancellation), the column sums ride along on
// the VALU.  The four waves take four contiguous row ranges and are combined through LDS in wave order; the result does
// not depend on the launch.  (The previous VALU version: one thread per column for the means, LDS-tiled pair products:
// 4.8 ms = 215 GB/s at C4's width.)
typedef double gd4_t __attribute__((ext_vector_type(4)));
template <int NCB>
__global__ __launch_bounds__(GT) void gelman_chain_mfma(const double* __restrict__ samples, long long S, int k, long long row0,
                                                        long long N, const int* __restrict__ cols, int p,
                                                        const double* __restrict__ center, double* __restrict__ work) {
  constexpr int NT2 = NCB * (NCB + 1) / 2;          // tile pairs (a-block <= b-block)
  __shared__ double s_acc[NT2 * 256];               // [tile][lane * 4 + r]
  __shared__ double s_sum[NCB * 16];                // column sums of d
  __shared__ double s_shift[NCB * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const long long c = blockIdx.x;
  const double* base = samples + c * (long long)k * S + row0;
  // this lane's column of every block (a padded column reads column 0 and is multiplied by 0)
  const double* colp[NCB];
  double shift[NCB], msk[NCB], csum[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) {
    const int a = cb * 16 + j;
    colp[cb] = base + (long long)cols[a < p ? a : 0] * S;
    shift[cb] = colp[cb][0];
    msk[cb] = a < p ? 1.0 : 0.0;
    csum[cb] = 0.0;
  }
  gd4_t acc[NT2];
#pragma unroll
  for (int t = 0; t < NT2; t++) acc[t] = (gd4_t){0.0, 0.0, 0.0, 0.0};
  // Rows of this wave: a contiguous quarter of the window, in groups of 16.  Lane (kk, j) loads rows 4 kk .. 4 kk + 3 of the
  // group for its column -- 32 contiguous bytes, the four kk classes together a full 128-byte line per column -- and K-step s
  // of the group multiplies rows {4 kk + s}: which four rows share a K-step only changes the order of a sum.  Two groups are
  // in flight behind the one being multiplied (the loads come from HBM: ~2 us, a group's 40 MFMAs are ~1 us).
  const long long groups = (N + 15) / 16, per = (groups + 3) / 4;
  const long long g_lo = wave * per, g_hi = (g_lo + per < groups) ? g_lo + per : groups;
  // (a pair is 8-byte aligned only: the window starts at any row of a history whose row stride may be odd -- the vector type
  //  says so, the loads are still one global_load_dwordx4 each, which gfx950 serves at any 4-byte alignment)
  typedef double gd2_t __attribute__((ext_vector_type(2), aligned(8)));
  double buf[3][NCB][4];
  auto load = [&](long long gi, double (&v)[NCB][4]) {
    const long long t = 16 * gi + 4 * kk;
    if (t + 3 < N) {
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        const gd2_t lo = *reinterpret_cast<const gd2_t*>(colp[cb] + t), hi = *reinterpret_cast<const gd2_t*>(colp[cb] + t + 2);
        v[cb][0] = lo[0]; v[cb][1] = lo[1]; v[cb][2] = hi[0]; v[cb][3] = hi[1];
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int u = 0; u < 4; u++) v[cb][u] = (t + u < N) ? colp[cb][t + u] : shift[cb];   // (a row beyond the window: d = 0)
    }
  };
  if (g_lo < g_hi) load(g_lo, buf[0]);
  if (g_lo + 1 < g_hi) load(g_lo + 1, buf[1]);
  for (long long gi = g_lo; gi < g_hi; gi += 3) {
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {       // buffer roles rotate statically
      if (gi + ph < g_hi) {
        if (gi + ph + 2 < g_hi) load(gi + ph + 2, buf[(ph + 2) % 3]);
        double (&cur)[NCB][4] = buf[ph];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          double d[NCB];
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) { d[cb] = (cur[cb][u] - shift[cb]) * msk[cb]; csum[cb] += d[cb]; }
          int t = 0;
#pragma unroll
          for (int ab = 0; ab < NCB; ab++)
#pragma unroll
            for (int bb = ab; bb < NCB; bb++, t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[ab], d[bb], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // column sums: the four kk classes of a column sit in lanes j, j + 16, j + 32, j + 48
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) {
    double v = csum[cb];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    csum[cb] = v;
  }
  // waves combine in wave order
  for (int w = 0; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          double* d = &s_acc[t * 256 + lane * 4 + r];
          *d = (w == 0) ? acc[t][r] : *d + acc[t][r];
        }
      if (lane < 16) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          double* d = &s_sum[cb * 16 + lane];
          *d = (w == 0) ? csum[cb] : *d + csum[cb];
          if (w == 0) s_shift[cb * 16 + lane] = shift[cb];
        }
      }
    }
    __syncthreads();
  }
  // finish: D register r of lane l is row 4 r + l / 16, column l % 16 of its tile
  double* wout = work + c * (long long)(p + p * p);
  const double dn = (double)N;
  if (tid < p) wout[tid] = (s_shift[tid] + s_sum[tid] / dn) - (center ? center[tid] : 0.0);
  for (int e = tid; e < NT2 * 256; e += GT) {
    const int t = e >> 8, q = e & 255, l = q >> 2, r = q & 3;
    int ab = 0, rem = t;
    while (rem >= NCB - ab) { rem -= NCB - ab; ab++; }
    const int bb = ab + rem;
    const int a = ab * 16 + 4 * r + (l >> 4), b2 = bb * 16 + (l & 15);
    if (a < p && b2 < p && (ab < bb || a <= b2)) {
      const double v = (s_acc[e] - s_sum[a] * s_sum[b2] / dn) / (dn - 1.0);
      wout[p + a * p + b2] = v;
      wout[p + b2 * p + a] = v;
    }
  }
}

// Fixed-order sum over the local chains -> partial.  64 elements of the partial per block, four threads per element: thread
// g sums the chains g, g + 4, g + 8, ... in chain order, the four sums are joined as ((g0 + g1) + g2) + g3.  (One block for
// everything -- 21 elements per thread, each a serial walk over 512 chains -- took 2.5 of the 3.0 ms of a check at C4.)
__global__ __launch_bounds__(GT) void gelman_sum_kernel(const double* __restrict__ work, long long C, int p,
                                                        double* __restrict__ partial) {
  __shared__ double s_g[4][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int stride = p + p * p;
  const int len = 1 + 5 * p + 2 * p * p;
  const int e = blockIdx.x * 64 + el;
  double s = 0.0;
  if (e < len) {
    if (e == 0) {
      s = (g == 0) ? (double)C : 0.0;
    } else if (e < 1 + p) {  // sum xbar
      int a = e - 1;
      for (long long c = g; c < C; c += 4) s += work[c * stride + a];
    } else if (e < 1 + p + p * p) {  // sum xbar xbar^T
      int ab = e - 1 - p, a = ab / p, b = ab % p;
      for (long long c = g; c < C; c += 4) s = __builtin_fma(work[c * stride + a], work[c * stride + b], s);
    } else if (e < 1 + p + 2 * p * p) {  // sum S_c
      int ab = e - 1 - p - p * p;
      for (long long c = g; c < C; c += 4) s += work[c * stride + p + ab];
    } else {
      int r = e - 1 - p - 2 * p * p, which = r / p, a = r % p;
      for (long long c = g; c < C; c += 4) {
        double s2 = work[c * stride + p + a * p + a], xb = work[c * stride + a];
        double v = (which == 0) ? s2 : (which == 1) ? s2 * s2 : (which == 2) ? s2 * xb : s2 * xb * xb;
        s += v;
      }
    }
  }
  s_g[g][el] = s;
  __syncthreads();
  if (g == 0 && e < len) partial[e] = ((s_g[0][el] + s_g[1][el]) + s_g[2][el]) + s_g[3][el];
}

// Largest eigenvalue of a symmetric matrix (row-major, destroyed): Householder reduction to tridiagonal form, then the QL
// iteration with implicit shifts on the diagonal / sub-diagonal pair (eigenvalues only, nothing accumulated): O(p^3) once and
// O(p^2) per eigenvalue.  (The cyclic Jacobi sweeps this replaces ran until the off-diagonal norm was below 1e-300:
// 1.3-3 ms per check at p = 50, more than the device reduction of a 1 GB window.)  NaN when the iteration does not converge.
double top_eigenvalue_sym(std::vector<double>& a, int n) {
  std::vector<double> d(n), e(n);
  if (n == 1) return a[0];
  for (int i = n - 1; i >= 1; i--) {          // reduce row i with a Householder reflector built from a[i][0..i-1]
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k2 = 0; k2 <= l; k2++) scale += fabs(a[i * n + k2]);
      if (scale == 0.0) {
        e[i] = a[i * n + l];
      } else {
        for (int k2 = 0; k2 <= l; k2++) {
          a[i * n + k2] /= scale;
          h += a[i * n + k2] * a[i * n + k2];
        }
        double f = a[i * n + l];
        const double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        a[i * n + l] = f - g;
        f = 0.0;
        for (int j = 0; j <= l; j++) {
          double gg = 0.0;
          for (int k2 = 0; k2 <= j; k2++) gg += a[j * n + k2] * a[i * n + k2];
          for (int k2 = j + 1; k2 <= l; k2++) gg += a[k2 * n + j] * a[i * n + k2];
          e[j] = gg / h;
          f += e[j] * a[i * n + j];
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; j++) {
          const double fj = a[i * n + j];
          const double gj = e[j] - hh * fj;
          e[j] = gj;
          for (int k2 = 0; k2 <= j; k2++) a[j * n + k2] -= fj * e[k2] + gj * a[i * n + k2];
        }
      }
    } else {
      e[i] = a[i * n + l];
    }
    d[i] = h;
  }
  e[0] = 0.0;
  for (int i = 0; i < n; i++) d[i] = a[i * n + i];
  // QL with implicit shifts
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; l++) {
    int iter = 0, m2;
    do {
      for (m2 = l; m2 < n - 1; m2++) {
        const double dd = fabs(d[m2]) + fabs(d[m2 + 1]);
        if (fabs(e[m2]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m2 != l) {
        if (iter++ == 60) return NAN;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m2] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
        double s2 = 1.0, c2 = 1.0, p2 = 0.0;
        int i;
        for (i = m2 - 1; i >= l; i--) {
          double f = s2 * e[i];
          const double b2 = c2 * e[i];
          r = hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p2;
            e[m2] = 0.0;
            break;
          }
          s2 = f / r;
          c2 = g / r;
          g = d[i + 1] - p2;
          r = (d[i] - g) * s2 + 2.0 * c2 * b2;
          p2 = s2 * r;
          d[i + 1] = g + p2;
          g = c2 * r - b2;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p2;
        e[l] = g;
        e[m2] = 0.0;
      }
    } while (m2 != l);
  }
  double emax = d[0];
  for (int i = 1; i < n; i++)
    if (d[i] > emax) emax = d[i];
  return emax;
}

}  // namespace

extern "C" {

int64_t fmcmc_gelman_partial_len(int32_t p) { return 1 + 5 * (int64_t)p + 2 * (int64_t)p * p; }
int64_t fmcmc_gelman_work_len(int64_t nchains, int32_t p) { return nchains * ((int64_t)p + (int64_t)p * p); }

int fmcmc_gelman_partial_dev(const double* samples, int64_t nchains, int32_t k, int64_t S, int64_t row0,
                             int64_t N, const int32_t* cols, int32_t p, const double* center,
                             double* work, double* partial, void* hip_stream) {
  if (!samples || !cols || !work || !partial || p < 1 || p > FMCMC_MAX_K_WAVE || nchains < 1 || N < 2 ||
      row0 < 0 || row0 + N > S)
    return FMCMC_ERR_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
#define GELMAN_LAUNCH(NCBV)                                                                                         \
  hipLaunchKernelGGL(gelman_chain_mfma<NCBV>, dim3((unsigned)nchains), dim3(GT), 0, st, samples, (long long)S, (int)k, \
                     (long long)row0, (long long)N, cols, (int)p, center, work)
  switch ((p + 15) / 16) {
    case 1: GELMAN_LAUNCH(1); break;
    case 2: GELMAN_LAUNCH(2); break;
    case 3: GELMAN_LAUNCH(3); break;
    default: GELMAN_LAUNCH(4); break;
  }
#undef GELMAN_LAUNCH
  const int plen = 1 + 5 * (int)p + 2 * (int)p * (int)p;
  hipLaunchKernelGGL(gelman_sum_kernel, dim3((unsigned)((plen + 63) / 64)), dim3(GT), 0, st, work, (long long)nchains, (int)p, partial);
  return hipGetLastError() == hipSuccess ? FMCMC_OK : FMCMC_ERR_DEVICE;
}

// Host finish. partial is the (all-reduced) vector; xbar sums are relative to `center`.
// psrf[p] point estimates (coda's "Point est."), *mpsrf multivariate (NaN when p == 1).
int fmcmc_gelman_finish(const double* P, int32_t p, int64_t N, double* psrf, double* mpsrf) {
  if (!P || p < 1 || p > FMCMC_MAX_K || N < 2) return FMCMC_ERR_ARG;   // (the host finish is general in p)
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
    const double emax = top_eigenvalue_sym(Z, p);
    if (!(emax == emax)) return FMCMC_ERR_CHAIN;
    *mpsrf = sqrt((1.0 - 1.0 / N) + (1.0 + 1.0 / p) * emax / N);
  }
  return FMCMC_OK;
}

}  // extern "C"
