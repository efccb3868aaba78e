#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  double v = 100.0 * (lane >> 4) + (lane & 15);
  out[lane] = dpp_d<0x152>(v);          // row_share:2 -> every lane reads lane 2 of its own row
  unsigned lo = (unsigned)lane;
  auto r = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  out[64 + lane] = (double)r[0];
  out[128 + lane] = (double)r[1];
}
int main() {
  double* d; hipMalloc(&d, 192 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int p = 0; p < 3; p++) { for (int i = 0; i < 64; i++) printf("%g ", h[64 * p + i]); printf("\n"); }
  return 0;
}
