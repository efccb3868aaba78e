// tools/probe_logit_grid.hip -- design probe for the logistic observation loop (config C5: n = 100,000, 5 covariates +
// intercept, 4 chains per 512-thread workgroup, 256 workgroups): time of ONE evaluation of the workgroup's chains when the
// per-observation term is  0.5 s eta - g(|eta|),  g(u) = log(2 cosh(u / 2)) = u / 2 + log1p(exp(-u)),  with g read off a
// per-row polynomial table in LDS (grid 1 / G, degree DEG), against the operation count of the round-2/3 routine
// (27 fp64 operations and 3 scattered LDS reads per observation and chain).  The linear part sum_i 0.5 s_i eta_i is a dot
// product of the coefficients with data-only column sums and never enters the loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGLOG2=3 -DDEG=8 tools/probe_logit_grid.hip -o probe
// Diagnostic only; not part of the product build.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#ifndef GLOG2
#define GLOG2 3
#endif
#ifndef DEG
#define DEG 8
#endif
#ifndef RED
#define RED 0      // 0: s = fract(u), j = (unsigned)u   1: shift trick (round to nearest), three adds
#endif
#ifndef UMAX
#define UMAX 64
#endif
#ifndef WIDE
#define WIDE 0     // 1: the table as (DEG + 1) / 2 arrays of 16-byte coefficient PAIRS, read with ds_read_b128
#endif
constexpr int G = 1 << GLOG2;
constexpr int NROWS = UMAX * G;
constexpr int STRIDE = NROWS + 1;   // (odd: a multiple of 64 lets the compiler fuse two reads into ds_read2st64_b64, half the LDS rate)
constexpr int NT = 512, CW = 4, PL = 5;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int MODE>
__global__ void __launch_bounds__(NT, 1) loop(const double* __restrict__ X, int n, const double* __restrict__ beta /*[chains][6], pre-scaled by G*/,
                                               const double* __restrict__ gtab /*[DEG+1][NROWS]*/, double* out, int reps) {
  extern __shared__ double s_tab[];
  for (int i = threadIdx.x; i < (DEG + 2) * STRIDE; i += NT) s_tab[i] = gtab[i];
  __syncthreads();
  typedef __attribute__((address_space(3))) const double* ldsc_t;
  typedef __attribute__((address_space(3))) const char* ldsb_t;
  typedef const double __attribute__((address_space(1))) * gptr_t;
  typedef const char __attribute__((address_space(1))) * gcptr_t;
  const int tid = threadIdx.x;
  const unsigned tabaddr = (unsigned)(unsigned long long)(ldsc_t)s_tab;
  gptr_t colp[PL];
#pragma unroll
  for (int u = 0; u < PL; u++) colp[u] = (gptr_t)(X + (size_t)u * n);
  auto ldg = [](gptr_t base, unsigned int boff) -> double { return *(gptr_t)((gcptr_t)base + boff); };
  const unsigned nn = (unsigned)n, blast = 8u * (nn - 1u);
  double total[CW] = {0, 0, 0, 0};
  for (int rep = 0; rep < reps; rep++) {
    double b0v[CW], bs[CW][PL];
#pragma unroll
    for (int c = 0; c < CW; c++) {
      const double* b = beta + ((size_t)(blockIdx.x * CW + c) * 6);
      const double jit = 1.0 + 1e-9 * rep;
      b0v[c] = b[0] * jit;
#pragma unroll
      for (int u = 0; u < PL; u++) bs[c][u] = __builtin_amdgcn_readfirstlane((int)0) + b[1 + u] * jit;   // (wave-uniform)
    }
    double acc[CW] = {0, 0, 0, 0};
    double xb0[PL], xb1[PL];
    unsigned boff = 8u * (unsigned)tid;
    {
      const unsigned b = boff < blast ? boff : blast;
#pragma unroll
      for (int u = 0; u < PL; u++) xb0[u] = ldg(colp[u], b);
      boff += 8u * NT;
      const unsigned b1 = boff < blast ? boff : blast;
#pragma unroll
      for (int u = 0; u < PL; u++) xb1[u] = ldg(colp[u], b1);
    }
    auto one = [&](double (&xb)[PL]) {
      double eta[CW];
#pragma unroll
      for (int c = 0; c < CW; c++) eta[c] = b0v[c];
#pragma unroll
      for (int u = 0; u < PL; u++) {
#pragma unroll
        for (int c = 0; c < CW; c++) eta[c] = __builtin_fma(xb[u], bs[c][u], eta[c]);
      }
      boff += 8u * NT;
      {
        const unsigned b = boff < blast ? boff : blast;
#pragma unroll
        for (int u = 0; u < PL; u++) xb[u] = ldg(colp[u], b);
      }
      double s[CW], q[CW], co[CW][DEG + 1];
#pragma unroll
      for (int c = 0; c < CW; c++) {
        const double ue = __builtin_fabs(eta[c]);
        unsigned j;
        if (RED == 0) {
          s[c] = __builtin_amdgcn_fract(ue);
          j = (unsigned)ue;
        } else {
          const double t = ue + 0x1.8p52;
          const double kd = t - 0x1.8p52;
          s[c] = ue - kd;
          j = (unsigned)(unsigned long long)__double_as_longlong(t);
        }
        if (WIDE) {
          typedef double v2d __attribute__((ext_vector_type(2)));
          typedef __attribute__((address_space(3))) const v2d* lds2_t;
          const ldsb_t row = (ldsb_t)(unsigned long long)(tabaddr + 16u * j);
#pragma unroll
          for (int k = 0; k <= DEG; k += 2) {
            const v2d pr = *(lds2_t)(row + 16 * STRIDE * (k / 2));
            co[c][k] = pr.x;
            if (k + 1 <= DEG) co[c][k + 1] = pr.y;
          }
        } else {
          const ldsb_t row = (ldsb_t)(unsigned long long)(tabaddr + 8u * j);
#pragma unroll
          for (int k = 0; k <= DEG; k++) co[c][k] = *(ldsc_t)(row + 8 * STRIDE * k);
        }
      }
      if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < CW; c++) q[c] = co[c][DEG];
#pragma unroll
        for (int k = DEG - 1; k >= 0; k--) {
#pragma unroll
          for (int c = 0; c < CW; c++) q[c] = __builtin_fma(s[c], q[c], co[c][k]);
        }
#pragma unroll
        for (int c = 0; c < CW; c++) acc[c] += q[c];
      } else {   // MODE 1: LDS reads only (polynomial replaced by one add per coefficient pair): what the reads alone cost
#pragma unroll
        for (int c = 0; c < CW; c++) { double t = s[c]; for (int k = 0; k <= DEG; k += 4) t += co[c][k]; acc[c] += t; }
      }
    };
    const unsigned T = (nn + NT - 1u) / NT;
    const bool last_valid = (unsigned)tid + NT * (T - 1u) < nn;
    unsigned it = 0;
    for (; it + 2u < T; it += 2u) { one(xb0); one(xb1); }
    if (T - it == 2u) { one(xb0); if (last_valid) one(xb1); }
    else if (T - it == 1u) { if (last_valid) one(xb0); }
#pragma unroll
    for (int c = 0; c < CW; c++) total[c] += wave_sum(acc[c]);
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int c = 0; c < CW; c++) out[(blockIdx.x * (NT / 64) + (tid >> 6)) * CW + c] = total[c];
  }
}

static long double gfun(long double u) { return 0.5L * u + log1pl(expl(-u)); }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 100000;
  const int reps = argc > 2 ? atoi(argv[2]) : 50;
  const int C = 1024, p = PL;
  std::vector<double> X((size_t)n * p), beta((size_t)C * 6);
  uint64_t s = 88172645463325252ull;
  auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  auto nrm = [&]() { double a = u01(), b = u01(); return sqrt(-2 * log(a + 1e-300)) * cos(6.283185307179586 * b); };
  const double b5[6] = {-1, .5, -.5, .25, -.25, 1.0};
  for (auto& v : X) v = nrm();
  for (int c = 0; c < C; c++) for (int j = 0; j < 6; j++) beta[c * 6 + j] = (b5[j] + 0.01 * nrm()) * G;
  // table: Chebyshev interpolation of g on [j / G, (j + 1) / G] in the variable s in [0, 1), long double
  std::vector<double> tab((size_t)(DEG + 2) * STRIDE, 0.0);
  auto tix = [](int k, int j) -> size_t { return WIDE ? ((size_t)(k / 2) * STRIDE + j) * 2 + (k & 1) : (size_t)k * STRIDE + j; };
  {
    const int d = DEG;
    for (int j = 0; j < NROWS; j++) {
      long double A[16][17];
      for (int i = 0; i <= d; i++) {
        const long double sx = 0.5L + 0.5L * cosl(3.14159265358979323846L * (2 * i + 1) / (2 * (d + 1)));
        long double pw = 1;
        for (int k = 0; k <= d; k++) { A[i][k] = pw; pw *= sx; }
        A[i][d + 1] = gfun(((long double)j + sx) / G);
      }
      for (int i = 0; i <= d; i++) {
        int piv = i;
        for (int r = i + 1; r <= d; r++) if (fabsl(A[r][i]) > fabsl(A[piv][i])) piv = r;
        for (int k = 0; k <= d + 1; k++) { long double t = A[i][k]; A[i][k] = A[piv][k]; A[piv][k] = t; }
        for (int r = 0; r <= d; r++) if (r != i) {
          const long double f = A[r][i] / A[i][i];
          for (int k = i; k <= d + 1; k++) A[r][k] -= f * A[i][k];
        }
      }
      for (int k = 0; k <= d; k++) tab[tix(k, j)] = (double)(A[k][d + 1] / A[k][k]);
    }
  }
  double *dX, *db, *dt, *dout;
  CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&db, beta.size() * 8)); CK(hipMalloc(&dt, tab.size() * 8));
  CK(hipMalloc(&dout, 256 * 8 * CW * 8));
  CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, beta.data(), beta.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  const size_t lds = tab.size() * 8;
  CK(hipFuncSetAttribute((const void*)loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++) {
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(loop<0>, dim3(256), dim3(NT), lds, 0, dX, n, db, dt, dout, reps);
      else hipLaunchKernelGGL(loop<1>, dim3(256), dim3(NT), lds, 0, dX, n, db, dt, dout, reps);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    std::vector<double> out(256 * 8 * CW);
    CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
    double got = 0; for (int w = 0; w < 8; w++) got += out[w * CW];        // chain 0 of workgroup 0, all reps
    // host reference of chain 0, rep 0 (long double), for the full-polynomial mode
    long double ref = 0;
    if (mode == 0) for (int i = 0; i < n; i++) {
      long double eta = beta[0] / G; for (int j = 0; j < p; j++) eta += (long double)beta[1 + j] / G * X[(size_t)j * n + i];
      ref += gfun(fabsl(eta));
    }
    printf("WIDE=%d G=%d DEG=%d RED=%d mode=%s: %.2f us per evaluation (4 chains x %d obs per workgroup, 256 workgroups); lds %zu B; chain0 sum/rep %.12g ref %.12Lg\n",
           WIDE, G, DEG, RED, mode ? "reads-only" : "full", best * 1e3 / reps, n, lds, got / reps, ref);
  }
  return 0;
}
