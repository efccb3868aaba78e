// tools/probe_logit_shard.hip -- design probe, second form: the logistic evaluation OBSERVATION-sharded (the decomposition of
// config C4's wide kernels): workgroup b of 256 owns the canonical lanes 2b, 2b + 1 -- a constant slice of 2 x ceil(n / 512)
// observations -- for ALL chains; thread = chain (NCH chains per thread side by side), the slice's covariates arrive as SCALAR
// operands (s_load from a compact copy), the coefficients of a chain sit in its lane's VGPRs.  Lanes of a wave then look up
// g(|eta|) for the SAME observation and neighbouring chains: their rows are neighbours or equal, so the scattered-row bank
// conflicts of the chain-sharded loop (3-way on the fine grid) are gone and the fine grid (1 / G, degree DEG) is affordable.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGLOG2=6 -DDEG=5 tools/probe_logit_shard.hip -o probe
// Diagnostic only; not part of the product build.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#ifndef GLOG2
#define GLOG2 6
#endif
#ifndef DEG
#define DEG 5
#endif
#ifndef UMAXX2
#define UMAXX2 75          // table covers u < UMAXX2 / 2
#endif
#ifndef NCH
#define NCH 2              // chains per thread
#endif
#ifndef XMODE
#define XMODE 0      // 0: covariates as scalar loads one pass ahead; 1: never reloaded (ablation); 2: two passes ahead; 3: from LDS
#endif
#ifndef SPREAD
#define SPREAD 0.01        // sd of the chains' coefficients around the centre
#endif
constexpr int G = 1 << GLOG2;
constexpr int NROWS = UMAXX2 * G / 2;
constexpr int STRIDE = NROWS + 1;
constexpr int NPAIR = (DEG + 2) / 2;
#ifndef NTHREADS
#define NTHREADS 512      // 1024: sixteen waves per CU (four per SIMD), one chain per thread
#endif
constexpr int NT = NTHREADS, PL = 5;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double s_of(double v) { return v; }

__global__ void __launch_bounds__(NT, 1) loop(const double* __restrict__ xs /*[256][nobs][PL]*/, int nobs, const double* __restrict__ beta /*[6][chains], x G*/,
                                               int nchains, const double* __restrict__ gtab, double* part /*[chains][512]*/, int reps) {
  extern __shared__ double s_tab[];
  for (int i = threadIdx.x; i < NPAIR * STRIDE * 2; i += NT) s_tab[i] = gtab[i];
  double* s_x = s_tab + NPAIR * STRIDE * 2;          // XMODE 3: this workgroup's slice, [nobs][6] doubles (48 B per observation)
#if XMODE == 3
  for (int i = threadIdx.x; i < nobs * 6; i += NT) { const int o = i / 6, u = i - 6 * o; s_x[i] = (u < PL) ? xs[((size_t)blockIdx.x * nobs + o) * PL + u] : 0.0; }
#endif
  __syncthreads();
  typedef __attribute__((address_space(3))) const v2d* lds2_t;
  typedef __attribute__((address_space(3))) const char* ldsb_t;
  typedef const double __attribute__((address_space(4))) * cptr_t;
  const int tid = threadIdx.x;
  const unsigned tabaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const double*)s_tab;
  const cptr_t slice = (cptr_t)(xs + (size_t)blockIdx.x * nobs * PL);
  for (int rep = 0; rep < reps; rep++) {
    for (int cb = 0; cb < nchains; cb += NT * NCH) {
      double b0[NCH], bs[NCH][PL], a0[NCH], a1[NCH];
#pragma unroll
      for (int h = 0; h < NCH; h++) {
        const int ch = (cb + h * NT + tid < nchains) ? cb + h * NT + tid : 0;
        const double jit = 1.0 + 1e-9 * rep;
        b0[h] = beta[ch] * jit;
#pragma unroll
        for (int u = 0; u < PL; u++) bs[h][u] = beta[(size_t)(1 + u) * nchains + ch] * jit;
        a0[h] = 0.0; a1[h] = 0.0;
      }
      // OB observations x NCH chains per pass, software-pipelined by hand: a pass issues the lookups of pass p + 1 (one burst of
      // ds_read_b128) and then runs the polynomials of pass p, whose coefficients were read a pass ago -- the LDS round trip hides
      // under a pass of arithmetic.  ONE wait per pass, at its top (lgkmcnt(0): scalar loads share the counter and return out of
      // order, so a counted wait is not available); the scalar loads of pass p + 2 are issued right behind it.
      constexpr int OB = 2;
      const int npass = nobs / OB;
      double xa[OB][PL], xb[OB][PL], sva[OB][NCH], svb[OB][NCH];
      v2d pra[OB][NCH][NPAIR], prb[OB][NCH][NPAIR];
      auto sload = [&](double (&x)[OB][PL], int pass) {
        int pc = pass < npass ? pass : npass - 1;
        if (XMODE == 4) pc &= 3;                                  // (ablation: the same four passes over and over -- always cache hits)
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
          for (int u = 0; u < PL; u++) x[b][u] = slice[(pc * OB + b) * PL + u];
      };
      auto front = [&](const double (&x)[OB][PL], double (&sv)[OB][NCH], v2d (&pr)[OB][NCH][NPAIR]) {   // eta, reduction, lookups
        unsigned ad[OB][NCH];
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
          for (int h = 0; h < NCH; h++) {
            double eta = b0[h];
#pragma unroll
            for (int u = 0; u < PL; u++) eta = __builtin_fma(x[b][u], bs[h][u], eta);
            const double ue = __builtin_fabs(eta);
            sv[b][h] = __builtin_amdgcn_fract(ue);
            ad[b][h] = tabaddr + 16u * (unsigned)ue;
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
          for (int h = 0; h < NCH; h++)
#pragma unroll
            for (int k = 0; k < NPAIR; k++) pr[b][h][k] = *(lds2_t)((ldsb_t)(unsigned long long)ad[b][h] + 16 * STRIDE * k);
#ifndef ILV
        __builtin_amdgcn_sched_barrier(0);
#endif
      };
      auto back = [&](const double (&sv)[OB][NCH], const v2d (&pr)[OB][NCH][NPAIR]) {                     // polynomials
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
          for (int h = 0; h < NCH; h++) {
            double co[2 * NPAIR];
#pragma unroll
            for (int k = 0; k < NPAIR; k++) { co[2 * k] = pr[b][h][k].x; co[2 * k + 1] = pr[b][h][k].y; }
            double q = co[DEG];
#pragma unroll
            for (int k = DEG - 1; k >= 0; k--) q = __builtin_fma(sv[b][h], q, co[k]);
            if (b & 1) a1[h] += q; else a0[h] += q;
          }
#ifdef ILV   // the lookups of the next pass (issued by front() just before) interleaved with this pass's polynomials: 1 read per ILV VALU
#pragma unroll
        for (int i = 0; i < OB * NCH * NPAIR; i++) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, ILV, 0); }
#endif
        __builtin_amdgcn_sched_barrier(0);
      };
#if 0
#define XMODE_UNUSED 0      // 0: covariates of pass p + 2 loaded one pass ahead (product); 1: never reloaded (timing ablation: what the
#endif
#if XMODE == 3
      // covariates from LDS: uniform-address ds_read_b128 (a broadcast), ONE register set re-loaded in place the moment a pass's
      // etas are formed; everything on lgkmcnt is then an LDS read, in order
      const unsigned xaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const double*)s_x;
      auto xload = [&](double (&x)[OB][PL], int pass) {
        const int pc = pass < npass ? pass : npass - 1;
        const ldsb_t base = (ldsb_t)(unsigned long long)(xaddr + (unsigned)(pc * OB) * 48u);
#pragma unroll
        for (int b = 0; b < OB; b++) {
          const v2d p0 = *(lds2_t)(base + 48 * b), p1 = *(lds2_t)(base + 48 * b + 16), p2 = *(lds2_t)(base + 48 * b + 32);
          x[b][0] = p0.x; x[b][1] = p0.y; x[b][2] = p1.x; x[b][3] = p1.y; x[b][4] = p2.x;
        }
      };
      auto front3 = [&](double (&x)[OB][PL], int nextpass, double (&sv)[OB][NCH], v2d (&pr)[OB][NCH][NPAIR]) {
        unsigned ad[OB][NCH];
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
          for (int h = 0; h < NCH; h++) {
            double eta = b0[h];
#pragma unroll
            for (int u = 0; u < PL; u++) eta = __builtin_fma(x[b][u], bs[h][u], eta);
            const double ue = __builtin_fabs(eta);
            sv[b][h] = __builtin_amdgcn_fract(ue);
            ad[b][h] = tabaddr + 16u * (unsigned)ue;
          }
        __builtin_amdgcn_sched_barrier(0);
        xload(x, nextpass);
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
          for (int h = 0; h < NCH; h++)
#pragma unroll
            for (int k = 0; k < NPAIR; k++) pr[b][h][k] = *(lds2_t)((ldsb_t)(unsigned long long)ad[b][h] + 16 * STRIDE * k);
        __builtin_amdgcn_sched_barrier(0);
      };
      xload(xa, 0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      front3(xa, 1, sva, pra);
      for (int p = 0; p < npass; p += 2) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        front3(xa, p + 2, svb, prb);              // pass p + 1
        back(sva, pra);                           // pass p
        __builtin_amdgcn_s_waitcnt(0xC07F);
        front3(xa, p + 3, sva, pra);              // pass p + 2
        if (p + 1 < npass) back(svb, prb);        // pass p + 1
      }
#elif XMODE == 2
      double xc[OB][PL];
      sload(xa, 0); sload(xb, 1); sload(xc, 2);
      front(xa, sva, pra);
      for (int p = 0; p < npass; p += 6) {       // (npass is a multiple of 2; the clamped repeats beyond the end are not accumulated)
        __builtin_amdgcn_s_waitcnt(0xC07F); sload(xa, p + 3); front(xb, svb, prb); if (p + 0 < npass) back(sva, pra);
        __builtin_amdgcn_s_waitcnt(0xC07F); sload(xb, p + 4); front(xc, sva, pra); if (p + 1 < npass) back(svb, prb);
        __builtin_amdgcn_s_waitcnt(0xC07F); sload(xc, p + 5); front(xa, svb, prb); if (p + 2 < npass) back(sva, pra);
        __builtin_amdgcn_s_waitcnt(0xC07F); sload(xa, p + 6); front(xb, sva, pra); if (p + 3 < npass) back(svb, prb);
        __builtin_amdgcn_s_waitcnt(0xC07F); sload(xb, p + 7); front(xc, svb, prb); if (p + 4 < npass) back(sva, pra);
        __builtin_amdgcn_s_waitcnt(0xC07F); sload(xc, p + 8); front(xa, sva, pra); if (p + 5 < npass) back(svb, prb);
      }
#else
      sload(xa, 0);
      sload(xb, 1);
      front(xa, sva, pra);                        // pass 0's lookups in flight
      for (int p = 0; p < npass; p += 2) {
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): pass p's coefficients, pass p + 1's covariates
        if (XMODE == 0 || XMODE == 4) sload(xa, p + 2);
        front(xb, svb, prb);                      // pass p + 1 (beyond the end: a clamped repeat, not accumulated)
        back(sva, pra);                           // pass p
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (XMODE == 0 || XMODE == 4) sload(xb, p + 3);
        front(xa, sva, pra);                      // pass p + 2
        if (p + 1 < npass) back(svb, prb);        // pass p + 1
      }
#endif
      __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
      for (int h = 0; h < NCH; h++) {
        const int ch = cb + h * NT + tid;
        v2d pr = {a0[h], a1[h]};
        if (ch < nchains) *(v2d*)(part + (size_t)ch * 512 + 2 * blockIdx.x) = pr;
      }
    }
  }
}

static long double gfun(long double u) { return 0.5L * u + log1pl(expl(-u)); }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 100000;
  const int reps = argc > 2 ? atoi(argv[2]) : 40;
  const int C = argc > 3 ? atoi(argv[3]) : 1024;
  const int nslots = (n + 511) / 512, nobs = 2 * nslots;
  std::vector<double> X((size_t)n * PL), beta((size_t)6 * C), xs((size_t)256 * nobs * PL, 0.0);
  uint64_t s = 88172645463325252ull;
  auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  auto nrm = [&]() { double a = u01(), b = u01(); return sqrt(-2 * log(a + 1e-300)) * cos(6.283185307179586 * b); };
  const double b5[6] = {-1, .5, -.5, .25, -.25, 1.0};
  for (auto& v : X) v = nrm();
  for (int c = 0; c < C; c++) for (int j = 0; j < 6; j++) beta[(size_t)j * C + c] = (b5[j] + SPREAD * nrm()) * G;
  // slice of workgroup b: observation o = 2 slot + q is canonical lane 2 b + q, slot `slot`: i = 512 slot + 2 b + q
  for (int b = 0; b < 256; b++)
    for (int o = 0; o < nobs; o++) {
      const long long i = 512ll * (o / 2) + 2 * b + (o % 2);
      for (int u = 0; u < PL; u++) xs[((size_t)b * nobs + o) * PL + u] = i < n ? X[(size_t)u * n + i] : 0.0;
    }
  std::vector<double> tab((size_t)NPAIR * STRIDE * 2, 0.0);
  auto tix = [](int k, int j) -> size_t { return ((size_t)(k / 2) * STRIDE + j) * 2 + (k & 1); };
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
  double *dxs, *db, *dt, *dpart;
  CK(hipMalloc(&dxs, xs.size() * 8)); CK(hipMalloc(&db, beta.size() * 8)); CK(hipMalloc(&dt, tab.size() * 8));
  CK(hipMalloc(&dpart, (size_t)C * 512 * 8));
  CK(hipMemcpy(dxs, xs.data(), xs.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, beta.data(), beta.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  const size_t lds = tab.size() * 8 + (XMODE == 3 ? (size_t)nobs * 6 * 8 : 0);
  CK(hipFuncSetAttribute((const void*)loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(loop, dim3(256), dim3(NT), lds, 0, dxs, nobs, db, C, dt, dpart, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<double> part((size_t)512);
  CK(hipMemcpy(part.data(), dpart, 512 * 8, hipMemcpyDeviceToHost));
  double got = 0; for (int l = 0; l < 512; l++) got += part[l];
  long double ref = 0;
  const double jit = 1.0 + 1e-9 * (reps - 1);
  for (int i = 0; i < n; i++) {
    long double eta = (long double)beta[0] * jit / G; for (int j = 0; j < PL; j++) eta += (long double)beta[(size_t)(1 + j) * C] * jit / G * X[(size_t)j * n + i];
    ref += gfun(fabsl(eta));
  }
  printf("shard G=%d DEG=%d NCH=%d spread=%g chains=%d: %.2f us per evaluation of all chains (n = %d); lds %zu B; chain0 sum %.12g ref %.12Lg\n",
         G, DEG, NCH, (double)SPREAD, C, best * 1e3 / reps, n, lds, got, ref);
  return 0;
}
