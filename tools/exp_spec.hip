// tools/exp_spec.hip -- standalone timing harness for mh_sweep_spec<3, 20, KIND> at config C3's shape (1024 chains,
// n = 10000, p = 3, kernel_adapt with warmup 500; -DEXP_KIND=4: kernel_ram).  Compiles in seconds (one kernel
// instantiation), so variants of the adaptive owner can be compared on one box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DEXP_KIND=4] [-DSPEC_STAMP] tools/exp_spec.hip -o exp
// Prints the average kernel time of a sweep, a checksum of the samples (variants that must not change results keep it) and,
// with -DSPEC_STAMP, the s_memtime shares of the owner's phases.  Diagnostic only; not part of the product build.
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
#include "../fmcmc_amd/csrc/mh_spec.hpp"
#ifndef EXP_KIND
#define EXP_KIND 3
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int C = 1024, n = 10000, p = 3, k = 5, kz = 5;
  const int nsteps = argc > 1 ? atoi(argv[1]) : 10000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<double> X((size_t)n * p), y(n), th((size_t)C * k), mu(k, 0.0), sc(k, 1.0), lb(k, -DBL_MAX), ub(k, DBL_MAX);
  uint64_t s = 88172645463325252ull;
  auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  auto nrm = [&]() { double a = u01(), b = u01(); return sqrt(-2 * log(a + 1e-300)) * cos(6.283185307179586 * b); };
  for (auto& v : X) v = nrm();
  for (int i = 0; i < n; i++) y[i] = 3 + 2 * X[i] - X[n + i] + 0.5 * X[2 * n + i] + 4 * nrm();
  for (int c = 0; c < C; c++) { for (int j = 0; j < 4; j++) th[c * k + j] = 0.1 * nrm(); th[c * k + 4] = fabs(4.5 + 0.1 * nrm()); }
  std::vector<uint8_t> fx(k, 0);
  const long long S = nsteps;
  double *dX, *dy, *dth, *dmu, *dsc, *dlb, *dub, *dsam, *dlp, *ddr, *df0, *dst_th, *ws;
  uint8_t* dfx; long long *dacc, *dss, *dabs; int *dstat, *dhave, *dnerr; unsigned* dbits; double *dSig, *dmean;
  CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dy, y.size() * 8)); CK(hipMalloc(&dth, th.size() * 8));
  CK(hipMalloc(&dmu, k * 8)); CK(hipMalloc(&dsc, k * 8)); CK(hipMalloc(&dlb, k * 8)); CK(hipMalloc(&dub, k * 8)); CK(hipMalloc(&dfx, k));
  CK(hipMalloc(&dsam, (size_t)C * k * S * 8)); CK(hipMalloc(&ddr, (size_t)C * k * S * 8)); CK(hipMalloc(&dlp, (size_t)C * S * 8));
  CK(hipMalloc(&df0, C * 8)); CK(hipMalloc(&dst_th, C * k * 8)); CK(hipMalloc(&dacc, C * 8)); CK(hipMalloc(&dss, C * 8));
  CK(hipMalloc(&dstat, C * 4)); CK(hipMalloc(&dbits, (size_t)C * ((nsteps + 31) / 32) * 4));
  CK(hipMalloc(&dabs, C * 8)); CK(hipMalloc(&dhave, C * 4)); CK(hipMalloc(&dnerr, C * 4)); CK(hipMalloc(&dSig, (size_t)C * k * k * 8)); CK(hipMalloc(&dmean, (size_t)C * k * 8));
  const size_t items = (size_t)C * nsteps;
  CK(hipMalloc(&ws, items * (kz + 1) * 8));
  CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu.data(), k * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), k * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlb, lb.data(), k * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dub, ub.data(), k * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfx, fx.data(), k, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, 0, 1215ull, 0ll, 0ll, (long long)C, (long long)nsteps, kz, (EXP_KIND == 4) ? (double)k : 0.0, ws, ws + items);
  SweepArgs A; memset(&A, 0, sizeof(A));
  A.family = FMCMC_FAM_GAUSSIAN_LINREG; A.p = p; A.intercept = 1; A.guard = 1; A.n = n; A.X = dX; A.y = dy;
  A.kind = EXP_KIND; A.k = k; A.scheme = FMCMC_SCHEME_JOINT; A.freq = 1;
  A.warmup = (EXP_KIND == 3) ? 500 : 0; A.until = INFINITY; A.eps = 1e-4; A.arate = 0.234; A.ram_df = (EXP_KIND == 4) ? (double)k : 0.0; A.ram_neg_exp = -2.0 / 3.0;
  A.mu = dmu; A.scale = dsc; A.lb = dlb; A.ub = dub; A.fixed = dfx;
  A.nchains = C; A.nsteps = nsteps; A.burnin = 0; A.thin = 1; A.S = S; A.ldS = S; A.bits_stride = (nsteps + 31) / 32; A.seed = 1215; A.rng_mode = FMCMC_RNG_FED; A.fresh = 1; A.kz = kz;
  A.fed_logu = ws; A.fed_z = ws + items;
  A.theta0 = dth; A.f0 = df0; A.samples = dsam; A.logpost = dlp; A.draws = ddr; A.accept_count = dacc; A.accept_bits = dbits;
  A.status = dstat; A.status_step = dss; A.status_theta = dst_th;
  A.abs_iter = dabs; A.Sigma = dSig; A.mean_prev = dmean; A.have_mean = dhave; A.nerrors = dnerr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mh_sweep_spec<3, 20, EXP_KIND>), dim3(C / 4), dim3(SPEC_NT), spec_lds_bytes(20, EXP_KIND >= 3), 0, A);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) tot += ms;
  }
  std::vector<double> sam((size_t)C * k * S); std::vector<long long> acc(C);
  CK(hipMemcpy(sam.data(), dsam, sam.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(acc.data(), dacc, C * 8, hipMemcpyDeviceToHost));
  uint64_t h = 1469598103934665603ull; for (double v : sam) { uint64_t b; memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; }
  long long na = 0; for (auto a : acc) na += a;
  const double ms = tot / reps;
  printf("spec KIND=%d: %.3f ms per %d-step sweep = %.1f ns/step -> %.3e samples/s; accept %.4f; checksum %016llx\n", EXP_KIND,
         ms, nsteps, ms * 1e6 / nsteps, (double)C * (nsteps - 1) / (ms * 1e-3), (double)na / ((double)C * (nsteps - 1)), (unsigned long long)h);
#ifdef SPEC_STAMP
  {  // owner-phase stamps (ticks per step, median over the owner waves), written over the head of the draws buffer
    std::vector<double> dr((size_t)C / 4 * 4 * 16);
    CK(hipMemcpy(dr.data(), ddr, dr.size() * 8, hipMemcpyDeviceToHost));
    const char* names[10] = {"wait for the partials", "fold + closed form", "decision", "recursive mean / cov", "factor", "propose + publish", "row stores", "prepare (log sigma, ...)", "-", "-"};
    for (int j = 0; j < 8; j++) {
      std::vector<double> col;
      for (size_t w = 0; w < (size_t)C; w++) col.push_back(dr[w * 16 + j] / dr[w * 16 + 15]);
      std::sort(col.begin(), col.end());
      printf("  %-28s %7.0f ticks per step\n", names[j], col[col.size() / 2]);
    }
  }
#endif
  return 0;
}
