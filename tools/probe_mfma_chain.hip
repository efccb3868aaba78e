// Probe: cycles per v_mfma_f64_16x16x4_f64 of ONE wave per SIMD as a function of the number of independent accumulators
// (dependent issue distance), register operands only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, double a, double b) {
  d4 c[NACC];
#pragma unroll
  for (int q = 0; q < NACC; q++) c[q] = (d4){(double)q, 1, 2, 3};
  double x = a + threadIdx.x * 1e-6, y = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < NACC; q++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[q], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int q = 0; q < NACC; q++) s += c[q][0] + c[q][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / ((double)iters * NACC);
  if (s == 12345.678) out[1] = s;
}
template <int NACC> void run(double* out, int bs) {
  k<NACC><<<256, bs>>>(out, 20000, 1.0000001, 1e-9); (void)hipDeviceSynchronize();
  double h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("block %4d  accumulators %d: %.1f s_memtime ticks per MFMA\n", bs, NACC, h);
}
int main() {
  double* out; (void)hipMalloc(&out, 64);
  for (int bs : {256, 512}) { run<1>(out, bs); run<2>(out, bs); run<3>(out, bs); run<4>(out, bs); run<6>(out, bs); }
  return 0;
}
