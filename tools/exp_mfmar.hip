// tools/exp_mfmar.hip -- standalone timing harness for mh_sweep_mfmar / mh_sweep_mfma at the headline shape
// (1024 chains, n = 10000, p = 3, kernel_normal).  Compiles in seconds (one kernel instantiation), so ablation builds
// (-DMFR_X=<bits>, see mh_mfma_rep.hpp) can be compared on one box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DMFR_X=n] [-DEXP_OWNERS] tools/exp_mfmar.hip -o exp
// Prints the average kernel time of a 10^4-step sweep and a checksum of the samples (ablations that must not change
// results keep the checksum).  Diagnostic only; not part of the product build.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>
#include <vector>
#include <algorithm>
#include "../include/fmcmc_amd.h"
#include "../include/fmh_detmath.h"
#include "../include/fmh_philox.h"
#include "../fmcmc_amd/csrc/mh_common.hpp"
#include "../fmcmc_amd/csrc/mh_pipe.hpp"
#include "../fmcmc_amd/csrc/mh_mfma.hpp"
#include "../fmcmc_amd/csrc/mh_mfma_rep.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int C = 1024, n = 10000, p = 3, k = 5, kz = 5;
  const int nsteps = argc > 1 ? atoi(argv[1]) : 10000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<double> X((size_t)n * p), y(n), th((size_t)C * k), mu(k, 0.0), sc(k, 0.02), lb(k, -DBL_MAX), ub(k, DBL_MAX);
  uint64_t s = 88172645463325252ull;
  auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  auto nrm = [&]() { double a = u01(), b = u01(); return sqrt(-2 * log(a + 1e-300)) * cos(6.283185307179586 * b); };
  for (auto& v : X) v = nrm();
  for (int i = 0; i < n; i++) y[i] = 3 + 2 * X[i] - X[n + i] + 0.5 * X[2 * n + i] + 4 * nrm();
  for (int c = 0; c < C; c++) { for (int j = 0; j < 4; j++) th[c * k + j] = 0.1 * nrm(); th[c * k + 4] = fabs(4.5 + 0.1 * nrm()); }
  std::vector<uint8_t> fx(k, 0);
  const long long S = nsteps;
  double *dX, *dy, *dth, *dmu, *dsc, *dlb, *dub, *dsam, *dlp, *ddr, *df0, *dst_th, *ws;
  uint8_t* dfx; long long *dacc, *dss; int* dstat; unsigned* dbits;
  CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dy, y.size() * 8)); CK(hipMalloc(&dth, th.size() * 8));
  CK(hipMalloc(&dmu, k * 8)); CK(hipMalloc(&dsc, k * 8)); CK(hipMalloc(&dlb, k * 8)); CK(hipMalloc(&dub, k * 8)); CK(hipMalloc(&dfx, k));
  CK(hipMalloc(&dsam, (size_t)C * k * S * 8)); CK(hipMalloc(&ddr, (size_t)C * k * S * 8)); CK(hipMalloc(&dlp, (size_t)C * S * 8));
  CK(hipMalloc(&df0, C * 8)); CK(hipMalloc(&dst_th, C * k * 8)); CK(hipMalloc(&dacc, C * 8)); CK(hipMalloc(&dss, C * 8));
  CK(hipMalloc(&dstat, C * 4)); CK(hipMalloc(&dbits, (size_t)C * ((nsteps + 31) / 32) * 4));
  const size_t items = (size_t)C * nsteps;
  CK(hipMalloc(&ws, items * (kz + 1) * 8));
  CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu.data(), k * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), k * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlb, lb.data(), k * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dub, ub.data(), k * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfx, fx.data(), k, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, 0, 1215ull, 0ll, 0ll, (long long)C, (long long)nsteps, kz, 0, ws, ws + items);
  SweepArgs A; memset(&A, 0, sizeof(A));
  A.family = FMCMC_FAM_GAUSSIAN_LINREG; A.p = p; A.intercept = 1; A.guard = 1; A.n = n; A.X = dX; A.y = dy;
  A.kind = FMCMC_KERNEL_NORMAL; A.k = k; A.scheme = FMCMC_SCHEME_JOINT; A.freq = 1;
  A.mu = dmu; A.scale = dsc; A.lb = dlb; A.ub = dub; A.fixed = dfx;
  A.nchains = C; A.nsteps = nsteps; A.burnin = 0; A.thin = 1; A.S = S; A.ldS = S; A.bits_stride = (nsteps + 31) / 32; A.seed = 1215; A.rng_mode = FMCMC_RNG_FED; A.fresh = 1; A.kz = kz;
  A.fed_logu = ws; A.fed_z = ws + items;
  A.theta0 = dth; A.f0 = df0; A.samples = dsam; A.logpost = dlp; A.draws = ddr; A.accept_count = dacc; A.accept_bits = dbits;
  A.status = dstat; A.status_step = dss; A.status_theta = dst_th;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
#ifdef EXP_OWNERS
#ifdef EXP_BIG
    hipLaunchKernelGGL((mh_sweep_mfma<1, 1, 20, false, true>), dim3(C / 4), dim3(NT), mfma_lds_bytes(), 0, A);
#else
    hipLaunchKernelGGL((mh_sweep_mfma<1, 1, 20, false>), dim3(C / 4), dim3(NT), mfma_lds_bytes(), 0, A);
#endif
#elif defined(EXP_DBG)
    hipLaunchKernelGGL((mh_sweep_mfmar<1, 1, 20, true>), dim3(C / 4), dim3(NT), mfmar_lds_bytes(), 0, A);
#else
    hipLaunchKernelGGL((mh_sweep_mfmar<1, 1, 20, false>), dim3(C / 4), dim3(NT), mfmar_lds_bytes(), 0, A);
#endif
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) tot += ms;
  }
  std::vector<double> sam((size_t)C * k * S); std::vector<long long> acc(C);
  CK(hipMemcpy(sam.data(), dsam, sam.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(acc.data(), dacc, C * 8, hipMemcpyDeviceToHost));
  uint64_t h = 1469598103934665603ull; for (double v : sam) { uint64_t b; memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; }
  long long na = 0; for (auto a : acc) na += a;
  const double ms = tot / reps;
  printf("%s MFR_X=%d: %.3f ms per %d-step sweep = %.1f ns/step -> %.3e samples/s; accept %.4f; checksum %016llx\n",
#ifdef EXP_OWNERS
         "owners",
#else
         "replicated",
#endif
#ifdef MFR_X
         MFR_X,
#else
         0,
#endif
         ms, nsteps, ms * 1e6 / nsteps, (double)C * (nsteps - 1) / (ms * 1e-3), (double)na / ((double)C * (nsteps - 1)), (unsigned long long)h);
#ifdef EXP_DBG
  {  // per-wave stamps (ticks per step, median over workgroups), written to the tail of the logpost buffer
    std::vector<double> lp((size_t)C * S);
    CK(hipMemcpy(lp.data(), dlp, lp.size() * 8, hipMemcpyDeviceToHost));
    printf("wave: eval | barrier wait | post phase || ticks from the barrier exit to: partials | total | ratio | decision (only the stamp selected by -DMFR_STAMP_SEL is live)\n");
    for (int w = 0; w < 8; w++) {
      std::vector<double> col[8];
      for (int b = 0; b < C / 4; b++) {
        const double* d = lp.data() + lp.size() - 8 * ((size_t)b * 8 + w + 1);
        for (int j = 0; j < 8; j++) col[j].push_back(d[j] / d[4]);
      }
      double m[8];
      for (int j = 0; j < 8; j++) { std::sort(col[j].begin(), col[j].end()); m[j] = col[j][col[j].size() / 2]; }
      printf("%d: %6.0f %6.0f %6.0f || %6.0f %6.0f %6.0f %6.0f\n", w, m[0], m[1], m[2], m[3], m[5], m[6], m[7]);
    }
  }
#endif
  return 0;
}
