// Standalone timing harness of shard_columns_mfma<2, 3> (the slice product of the wide sweeps, mh_common.hpp) at config C4's
// shape: 256 workgroups x 8 waves, p = 48 (12 K-blocks), 3 M-tiles, one chain group of 256 chains = 16 N-tiles per visit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/exp tools/exp_shard_mfma.hip && /tmp/exp
//   -DEXP_T10: the form C4 runs (third M-tile as two 4x4x4 MFMAs per K-block)   -DEXP_W12: 12 waves, modes 7 / 8 = three per SIMD
//   (the all-4x4x4 form "T4" of profiles/r03_exp_shard_t10.txt was dropped with its code)
// mode 0: the evaluator waves of mh_sweep_wide2 (2..7) with their tile shares, waves 0, 1 idle
// mode 1: ONE wave per SIMD (4..7 -> 4 tiles each), the others idle           mode 2: waves 4, 5 alone (4 tiles each)
// mode 3: as mode 0 with waves 0, 1 running dependent fp64 FMAs (an owner's arithmetic on the same SIMDs)   mode 4: those FMAs alone
// (1500 dependent FMAs per "visit": w0 / 1500 = time per dependent fp64 instruction of an owner)
// Prints the time of one visit (us) per wave, median over workgroups; the B operands come from an L2-resident table here
// (in the sweep they were written by other XCDs a hand-over ago and miss).
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
#include <type_traits>
#include "../include/fmcmc_amd.h"
#include "../include/fmh_detmath.h"
#include "../include/fmh_philox.h"
#include "../fmcmc_amd/csrc/mh_common.hpp"
#ifndef EXP_KBC
#define EXP_KBC 12
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
namespace {
#ifdef EXP_W12
#define EXP_NT 768
#else
#define EXP_NT NT
#endif
__global__ __launch_bounds__(EXP_NT) void bench(const double* th, double* part, double* out, int reps, int mode, int p, int nchains) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KB = (p + 3) >> 2, mblk = shm_hdr(3) + 3 * KB * 64;
  for (int i = tid; i < mblk; i += NT) smem[i] = (i < 32) ? 0.0 : 1e-3 * (double)((i * 2654435761u) >> 20);
  if (tid < 64) ((unsigned*)smem)[tid] = 0xfffu;
  __syncthreads();
  ShardMfma sm;
  sm.th = th; sm.part = part; sm.p = p; sm.ic = 1; sm.lane0 = (int)blockIdx.x * 2; sm.tcount = 0;
  sm.lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const double*)smem;
  sm.NC = nchains; sm.ncp = 2 * nchains + SH_PAD; sm.cstride = 2; sm.coff = 0; sm.thoff = 0;
  bool active = false;
  if (mode == 0 || mode == 3) {
    active = wave >= 2;
    sm.tfirst = (wave == 4) ? 0 : (wave == 5) ? 1 : (wave == 2) ? 2 : (wave == 6) ? 6 : (wave == 3) ? 3 : 7;
    sm.tstep = (wave == 4 || wave == 5) ? 4 : 8;
  } else if (mode == 1) {
    active = wave >= 4; sm.tfirst = wave - 4; sm.tstep = 4;
  } else if (mode == 2) {
    active = wave == 4 || wave == 5; sm.tfirst = wave - 4; sm.tstep = 4;
  } else if (mode == 5 || mode == 6) {   // the product's shares 3 | 3 | 3 + 2 | 3 + 2 (8 waves; 6: owners' FMAs beside them)
    active = wave >= 2 && wave < 8;
    sm.tfirst = (wave == 4) ? 0 : (wave == 5) ? 3 : (wave == 2) ? 6 : (wave == 6) ? 7 : (wave == 3) ? 11 : 12;
    sm.tstep = (wave == 4 || wave == 5) ? 1 : 2;
    sm.tcount = (wave == 6 || wave == 7) ? 2 : 3;
  } else if (mode == 7 || mode == 8) {   // (-DEXP_W12) three waves per SIMD: 2 + 1 | 2 + 1 | 2 + 2 + 1 | 2 + 2 + 1 (8: owners' FMAs)
    active = wave >= 2;
    const int simd = wave & 3, idx = (wave >> 2) - (simd < 2 ? 1 : 0);      // idx: 0, 1(, 2) among the SIMD's evaluators
    const int base = (simd == 0) ? 0 : (simd == 1) ? 3 : (simd == 2) ? 6 : 11;
    const int nev = (simd < 2) ? 2 : 3;
    sm.tfirst = base + idx; sm.tstep = nev;
    sm.tcount = (simd < 2) ? (idx == 0 ? 2 : 1) : (idx < 2 ? 2 : 1);
  } else {
    active = false; sm.tfirst = 0; sm.tstep = 4;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (active) {
    for (int r = 0; r < reps; r++) {
#if defined(EXP_T10)
      shard_columns_mfma<2, 3, EXP_KBC, true>(sm);
#else
      shard_columns_mfma<2, 3, EXP_KBC>(sm);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if ((mode == 3 || mode == 4 || mode == 6 || mode == 8) && wave < 2) {
    double a = 1.0 + lane * 1e-9, b = 0.999999;
    for (int r = 0; r < reps * 25; r++) {
#pragma unroll
      for (int u = 0; u < 60; u++) a = fmh_fma(a, b, 1e-9);
    }
    if (a == 0.123) part[0] = a;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(long long)blockIdx.x * 12 + wave] = (double)(t1 - t0) / reps;
}
}  // namespace
int main() {
  const int p = 48, ng = 256, reps = 200;
  const size_t nth = (size_t)(p + 1) * (2 * ng + SH_PAD), npart = (size_t)(2 * ng) * (NT + SH_PAD);
  double *th, *part, *out;
  CK(hipMalloc(&th, nth * 8)); CK(hipMalloc(&part, npart * 8)); CK(hipMalloc(&out, 256 * 12 * 8));
  std::vector<double> h(nth);
  for (size_t i = 0; i < nth; i++) h[i] = 1e-3 * (double)(i % 977);
  CK(hipMemcpy(th, h.data(), nth * 8, hipMemcpyHostToDevice));
  const int KB = (p + 3) / 4; const size_t lds = sizeof(double) * (shm_hdr(3) + 3 * KB * 64);
  for (int mode = 0; mode < 9; mode++) {
#ifndef EXP_W12
    if (mode >= 7) break;
#endif
    for (int w = 0; w < 2; w++) {
      hipLaunchKernelGGL(bench, dim3(256), dim3(EXP_NT), lds, 0, th, part, out, reps, mode, p, ng);
      CK(hipDeviceSynchronize());
    }
    std::vector<double> o(256 * 12);
    CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
    printf("mode %d: visit time per wave (us, s_memtime at ~2330 ticks per us):", mode);
    for (int w = 0; w < EXP_NT / 64; w++) {
      std::vector<double> v;
      for (int b = 0; b < 256; b++) v.push_back(o[b * 12 + w]);
      std::sort(v.begin(), v.end());
      printf("  w%d %.2f", w, v[128] / 2330.0);
    }
    printf("\n");
  }
  return 0;
}
