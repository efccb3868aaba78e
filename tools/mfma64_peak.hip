// Microbenchmark: v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 rates, alone and beside VALU fp64 FMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_ __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: 16x16x4 only, 1: 4x4x4 only, 2: 16x16x4 + VALU fma, 3: VALU only
__global__ void k(double* out, int iters, double a, double b) {
  double4_ c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
  double d0 = 0, d1 = 1, d2 = 2, d3 = 3;
  double v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = threadIdx.x * 1e-3 + j;
  double x = a + threadIdx.x * 1e-6, y = b;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0 || MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0);
    }
    if (MODE == 1) {
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, d3, 0, 0, 0);
    }
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = __builtin_fma(v[j], x, y);
    }
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3] + d0 + d1 + d2 + d3;
#pragma unroll
  for (int j = 0; j < 8; j++) s += v[j];
  if (s == 12345.678) out[threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int bs, double* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000, blocks = 256;
  k<MODE><<<blocks, bs>>>(out, 100, 1.0000001, 1e-9); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<MODE><<<blocks, bs>>>(out, iters, 1.0000001, 1e-9); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double waves = bs / 64.0 * blocks;
  double mfma_flops = (MODE == 0 || MODE == 2) ? 4.0 * 2048 : (MODE == 1 ? 4.0 * 512 : 0);
  double valu_flops = (MODE >= 2) ? 32.0 * 2 * 64 : 0;
  printf("%-22s block=%4d: %.3f ms  mfma %.2f TF + valu %.2f TF ; ns per loop iter per wave %.1f\n", name, bs, ms,
         mfma_flops * iters * waves / ms / 1e9, valu_flops * iters * waves / ms / 1e9, ms * 1e6 / iters);
}
int main() {
  double* out; (void)hipMalloc(&out, 1 << 20);
  for (int bs : {256, 512}) {
    if (bs == 256) { run<0>("mfma16x16x4", 256, out); run<1>("mfma4x4x4_4b", 256, out); run<2>("mfma16 + 32 valu fma", 256, out); run<3>("32 valu fma", 256, out); }
    else { run<0>("mfma16x16x4", 512, out); run<1>("mfma4x4x4_4b", 512, out); run<2>("mfma16 + 32 valu fma", 512, out); run<3>("32 valu fma", 512, out); }
  }
  return 0;
}
