// tools/scalar_slice_probe.hip -- can a 15 KB per-workgroup slice of X be consumed as SCALAR operands every MH step?
// Each workgroup (8 waves) re-reads its own contiguous 15 KB slice with uniform (scalar) loads and feeds 40 FMAs per
// column into per-thread accumulators -- the inner loop of an observation-sharded evaluation: 49 columns
// x 40 observations, thread = chain.  Reports the time per "step".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int NOBS = 40, NCOL = 49;
__global__ __launch_bounds__(512) void k(const double* __restrict__ Xs, const double* __restrict__ TH, double* out, int steps, int nchains) {
  const double* xs = Xs + (long long)blockIdx.x * NOBS * NCOL;      // this workgroup's slice, [col][obs]
  const int chain = threadIdx.x % nchains;
  double tot = 0.0;
  for (int s = 0; s < steps; s++) {
#ifdef WITH_FENCE   /* what an agent-scope acquire per step (the grid barrier's) does to the scalar loads */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    double mu[NOBS];
#pragma unroll
    for (int o = 0; o < NOBS; o++) mu[o] = 0.0;
    for (int j = 0; j < NCOL; j++) {
      const double b = TH[((long long)(s & 1) * NCOL + j) * nchains + chain];      // coefficient of this thread's chain
#pragma unroll
      for (int o = 0; o < NOBS; o++) mu[o] = __builtin_fma(xs[j * NOBS + o], b, mu[o]);   // uniform address: scalar load
    }
#pragma unroll
    for (int o = 0; o < NOBS; o++) tot += mu[o] * mu[o];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}
int main() {
  const int G = 256, C = 512, steps = 200;
  double *Xs, *TH, *out;
  hipMalloc(&Xs, (size_t)G * NOBS * NCOL * 8); hipMalloc(&TH, (size_t)2 * NCOL * C * 8); hipMalloc(&out, (size_t)G * 512 * 8);
  {  // real-looking operands (all-zero data ran the same loop at the same speed: no data-dependent clock effect)
    size_t nx = (size_t)G * NOBS * NCOL, nt = (size_t)2 * NCOL * C;
    double* h = (double*)malloc((nx > nt ? nx : nt) * 8);
    unsigned long long r = 88172645463325252ull;
    for (size_t i = 0; i < nx; i++) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; h[i] = (double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
    hipMemcpy(Xs, h, nx * 8, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nt; i++) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; h[i] = (double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
    hipMemcpy(TH, h, nt * 8, hipMemcpyHostToDevice);
    free(h);
  }
  for (int blocks : {8, 64, 256}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, Xs, TH, out, steps, C);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %3d: %.2f us per step (49 columns x 40 observations x 512 chains per workgroup)\n", blocks, ms * 1e3 / steps);
  }
  return 0;
}
