// probe_eval_rate.hip -- what one fp64 VALU instruction of the linear model's evaluation loop costs a SIMD (round 5).
// The loop of mh_sweep_lat / mh_sweep_spec: per observation slot  m = fma(x1, b1, b0); m = fma(x2, b2, m); m = fma(x3, b3, m);
// r = y - m; a = fma(r, r, a)  with 20 slots of distinct x / y registers per lane (160 VGPRs of data).
//   V=0  the loop as the kernels have it             V=1  every fma reads the SAME x register (operand fetch / bank effects off)
//   V=2  coefficients in SGPRs                        V=3  slots in two halves of 10 (80 data registers: half the operands)
// Threads per workgroup TPW = 512 (2 waves per SIMD) or 768 (3, with 12 slots); one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DV=0 -DTPW=512 tools/probe_eval_rate.hip -o tools/exp_bin/probe_eval_rate_v0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef V
#define V 0
#endif
#ifndef TPW
#define TPW 512
#endif
#ifndef OPT
#define OPT (TPW == 512 ? 20 : 12)
#endif
__global__ __launch_bounds__(TPW) void k(const double* X, const double* y, const double* th, double* out, int iters) {
  const int tid = threadIdx.x;
  double xr[OPT][3], yr[OPT];
#pragma unroll
  for (int s = 0; s < OPT; s++) {
#pragma unroll
    for (int j = 0; j < 3; j++) xr[s][j] = X[(size_t)j * TPW * OPT + s * TPW + tid];
    yr[s] = y[s * TPW + tid];
  }
  double b0 = th[0], b1 = th[1], b2 = th[2], b3 = th[3];
  double tot = 0.0;
  for (int it = 0; it < iters; it++) {
#if V != 2
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
#else
    asm volatile("" : "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3));
#endif
    double a = 0.0;
#pragma unroll
    for (int s = 0; s < OPT; s++) {
#if V == 1
      double m = __builtin_fma(xr[0][0], b1, b0);
      m = __builtin_fma(xr[0][0], b2, m);
      m = __builtin_fma(xr[0][0], b3, m);
      const double r = yr[0] - m;
#else
      double m = __builtin_fma(xr[s][0], b1, b0);
      m = __builtin_fma(xr[s][1], b2, m);
      m = __builtin_fma(xr[s][2], b3, m);
      const double r = yr[s] - m;
#endif
      a = __builtin_fma(r, r, a);
    }
    tot += a;
    b0 += 1e-9 * tot;   // (a dependence from one evaluation to the next, as the sweep has)
  }
  // keep every data register alive
  double keep = 0.0;
#pragma unroll
  for (int s = 0; s < OPT; s++) keep += xr[s][0] + xr[s][1] + xr[s][2] + yr[s];
  out[blockIdx.x * TPW + tid] = tot + 1e-300 * keep;
}
int main() {
  const int nb = 256, iters = 20000;
  std::vector<double> hx((size_t)3 * TPW * OPT, 0.5), hy((size_t)TPW * OPT, 1.0), ht = {0.1, 0.2, 0.3, 0.4};
  double *X, *y, *th, *out;
  hipMalloc(&X, hx.size() * 8); hipMalloc(&y, hy.size() * 8); hipMalloc(&th, 32); hipMalloc(&out, (size_t)nb * TPW * 8);
  hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(th, ht.data(), 32, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {4, 256}) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(TPW), 0, 0, X, y, th, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(TPW), 0, 0, X, y, th, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_eval = ms * 1e6 / iters, instr_simd = 5.0 * OPT * (TPW / 64) / 4.0;
    printf("V=%d TPW=%d OPT=%d grid=%d: %.1f ns per evaluation, %.2f ns = %.2f cycles (2.4 GHz) per fp64 instruction and SIMD\n", V, TPW, OPT, grid,
           ns_eval, ns_eval / instr_simd, ns_eval / instr_simd * 2.4);
  }
  return 0;
}
