// Microbenchmark: sustained v_fma_f64 rate per CU (1, 2, 4 waves per SIMD), to calibrate the roofline.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int ILP>
__global__ void fma_loop(double* out, int iters, double a, double b) {
  double acc[ILP];
#pragma unroll
  for (int j = 0; j < ILP; j++) acc[j] = threadIdx.x * 1e-3 + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = __builtin_fma(acc[j], a, b);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < ILP; j++) s += acc[j];
  if (s == 12345.678) out[threadIdx.x] = s;
}
int main() {
  double* out; hipMalloc(&out, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int bs : {256, 512, 1024}) {
    for (int blocks : {256, 512}) {
      fma_loop<16><<<blocks, bs>>>(out, 100, 1.0000001, 1e-9);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      fma_loop<16><<<blocks, bs>>>(out, iters, 1.0000001, 1e-9);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = 2.0 * 16 * iters * (double)bs * blocks;
      printf("block=%4d blocks=%3d : %.3f ms  %.2f TFLOP/s fp64 ; cycles/fma-instr/SIMD (at 2.4GHz, if 1 WG/CU) = %.2f\n", bs, blocks, ms,
             flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (16.0 * iters * (bs / 64.0) / 4.0 * (blocks / 256.0)));
    }
  }
  return 0;
}
