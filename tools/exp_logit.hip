// tools/exp_logit.hip -- standalone timing harness for the streamed kernel on config C5 (logistic regression n = 100,000,
// k = 6, kernel_normal_reflective, 1024 chains per GPU): compares chains per workgroup (CW) and workgroups per CU (MINB).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCWV=2 -DMINBV=2 -DFAMV=2 tools/exp_logit.hip -o exp
// Diagnostic only; not part of the product build.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>
#include <vector>
#include <type_traits>
#include "../include/fmcmc_amd.h"
#include "../include/fmh_detmath.h"
#include "../include/fmh_philox.h"
#include "../fmcmc_amd/csrc/mh_common.hpp"
#include "../fmcmc_amd/csrc/mh_streamed.hpp"
#ifndef CWV
#define CWV 4
#endif
#ifndef MINBV
#define MINBV 1
#endif
#ifndef FAMV
#define FAMV 0
#endif
#ifndef KINDV
#define KINDV 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int C = 1024, p = 5, k = 6, kz = 6;
  const int n = argc > 3 ? atoi(argv[3]) : 100000;
  const int nsteps = argc > 1 ? atoi(argv[1]) : 200;
  const int reps = argc > 2 ? atoi(argv[2]) : 2;
  const int thin = 10;
  std::vector<double> X((size_t)n * p), y(n), th((size_t)C * k), mu(k, 0.0), sc(k, 0.01), lb(k, -5.0), ub(k, 5.0);
  uint64_t s = 88172645463325252ull;
  auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  auto nrm = [&]() { double a = u01(), b = u01(); return sqrt(-2 * log(a + 1e-300)) * cos(6.283185307179586 * b); };
  const double b5[6] = {-1, .5, -.5, .25, -.25, 1.0};
  for (auto& v : X) v = nrm();
  for (int i = 0; i < n; i++) {
    double eta = b5[0];
    for (int j = 0; j < p; j++) eta += b5[j + 1] * X[(size_t)j * n + i];
    y[i] = (u01() < 1 / (1 + exp(-eta))) ? 1.0 : 0.0;
  }
  for (int c = 0; c < C; c++) for (int j = 0; j < k; j++) th[c * k + j] = b5[j] + 0.01 * nrm();
  std::vector<uint8_t> fx(k, 0);
  const long long S = nsteps / thin;
  double *dX, *dy, *dth, *dmu, *dsc, *dlb, *dub, *dsam, *dlp, *ddr, *df0, *dst_th;
  uint8_t* dfx; long long *dacc, *dss; int* dstat; unsigned* dbits;
  CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dy, y.size() * 8)); CK(hipMalloc(&dth, th.size() * 8));
  CK(hipMalloc(&dmu, k * 8)); CK(hipMalloc(&dsc, k * 8)); CK(hipMalloc(&dlb, k * 8)); CK(hipMalloc(&dub, k * 8)); CK(hipMalloc(&dfx, k));
  CK(hipMalloc(&dsam, (size_t)C * k * S * 8)); CK(hipMalloc(&ddr, (size_t)C * k * S * 8)); CK(hipMalloc(&dlp, (size_t)C * S * 8));
  CK(hipMalloc(&df0, C * 8)); CK(hipMalloc(&dst_th, C * k * 8)); CK(hipMalloc(&dacc, C * 8)); CK(hipMalloc(&dss, C * 8));
  CK(hipMalloc(&dstat, C * 4)); CK(hipMalloc(&dbits, (size_t)C * ((nsteps + 31) / 32) * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu.data(), k * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), k * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlb, lb.data(), k * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dub, ub.data(), k * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfx, fx.data(), k, hipMemcpyHostToDevice));
  SweepArgs A; memset(&A, 0, sizeof(A));
  A.family = FMCMC_FAM_LOGISTIC; A.p = p; A.intercept = 1; A.guard = 0; A.n = n; A.X = dX; A.y = dy; A.prior_div = 8.0;
  A.kind = FMCMC_KERNEL_NORMAL_REFLECTIVE; A.k = k; A.scheme = FMCMC_SCHEME_JOINT; A.freq = 1;
  A.mu = dmu; A.scale = dsc; A.lb = dlb; A.ub = dub; A.fixed = dfx;
  A.nchains = C; A.nsteps = nsteps; A.burnin = 0; A.thin = thin; A.S = S; A.ldS = S; A.seed = 1215; A.rng_mode = FMCMC_RNG_PHILOX; A.fresh = 1; A.kz = kz;
  A.tb = 32;
  A.theta0 = dth; A.f0 = df0; A.samples = dsam; A.logpost = dlp; A.draws = ddr; A.accept_count = dacc; A.accept_bits = dbits;
  A.status = dstat; A.status_step = dss; A.status_theta = dst_th;
  const int kf = k;
  const size_t ldsd = 4 * (size_t)k + (k / 2 + 1) + (size_t)NW * CWV + 1 + (size_t)CWV * A.tb * (kz + 1) + (size_t)CWV * chain_lds_doubles(k, kf, A.kind);
  const size_t lds = (ldsd + (FAMV == 2 ? SP_LDS_DOUBLES + 1 : 0)) * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mh_sweep_kernel<CWV, -1, 0, KINDV, FAMV, MINBV>), dim3(C / CWV), dim3(NT), lds, 0, A);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) tot += ms;
  }
  std::vector<double> sam((size_t)C * k * S); std::vector<long long> acc(C);
  CK(hipMemcpy(sam.data(), dsam, sam.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(acc.data(), dacc, C * 8, hipMemcpyDeviceToHost));
  uint64_t h = 1469598103934665603ull; for (double v : sam) { uint64_t b; memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; }
  long long na = 0; for (auto a : acc) na += a;
  const double ms = tot / reps;
  printf("CW=%d MINB=%d FAM=%d KIND=%d lds=%zu: %.2f ms per %d steps = %.1f us/step -> %.3e samples/s; accept %.4f; checksum %016llx\n",
         CWV, MINBV, FAMV, KINDV, lds, ms, nsteps, ms * 1e3 / nsteps, (double)C * (nsteps - 1) / (ms * 1e-3), (double)na / ((double)C * (nsteps - 1)), (unsigned long long)h);
  return 0;
}
