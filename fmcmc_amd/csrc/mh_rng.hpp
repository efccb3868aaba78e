// mh_rng.hpp -- rng_fill_kernel (canonical Philox stream -> HBM, compiled into the engine translation unit only) and the
// constants / stamp helper shared by the stream-fed kernels (mh_mfma.hpp, mh_spec.hpp, mh_mfma_ad.hpp).
#pragma once

namespace {

constexpr int PIPE_KMAX = 16; // parameters per chain supported by the stream-fed kernels
constexpr int PIPE_TRS = 66;  // row stride (doubles) of the transposed lane-partial tile

#ifdef FMH_WITH_RNG_FILL
// Canonical Philox stream materialised in HBM for the pipelined kernel (same layout as FED mode):
// logu[c][i-1] = log accept-uniform of loop step i, z[c][i-1][a] = a-th proposal variate of step i.
// Keeping Philox + AS241 (about 50 fp64 constants) out of the sweep kernel leaves its VGPR file
// to the observation data.  48 B per chain-step at k = 5: noise next to the 8 TB/s of HBM.
__global__ __launch_bounds__(256) void rng_fill_kernel(unsigned long long seed, long long step_base,
                                                       long long chain_base, long long nchains,
                                                       long long nsteps, int kz, double student_df,
                                                       double* __restrict__ logu, double* __restrict__ z) {
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= nchains * nsteps) return;
  const long long c = item / nsteps;
  const long long i = item - c * nsteps + 1;  // R's 1-based loop index
  const unsigned int st = (unsigned int)(step_base + i), cg = (unsigned int)(chain_base + c);
  if (i < 2) {  // row 1 draws nothing
    logu[item] = 0.0;
    for (int a = 0; a < kz; a++) z[item * kz + a] = 0.0;
    return;
  }
  logu[item] = fmh_log_accept_u(seed, st, cg);
  if (student_df > 0) {  // kernel_ram: qfun = rt(k, df)
    for (int a = 0; a < kz; a++) z[item * kz + a] = fmh_student_t(seed, st, cg, (unsigned int)a, student_df);
    return;
  }
  if (student_df < 0) {  // uniform kernels: the unif_rand() behind runif (R/kernel_unif.R:74)
    for (int a = 0; a < kz; a++) z[item * kz + a] = fmh_unif(seed, st, cg, (unsigned int)a);
    return;
  }
  for (int b = 0; 2 * b < kz; b++) {
    double u0, u1;
    fmh_uniform2(seed, st, cg, (unsigned int)b, FMH_STREAM_NORMAL, &u0, &u1);
    z[item * kz + 2 * b] = fmh_qnorm(u0);
    if (2 * b + 1 < kz) z[item * kz + 2 * b + 1] = fmh_qnorm(u1);
  }
}
#endif  // FMH_WITH_RNG_FILL

__device__ __forceinline__ unsigned long long clk() {  // diagnostic stamp (debug mode 8 only)
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

}  // namespace
