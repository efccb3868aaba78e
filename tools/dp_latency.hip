// tools/dp_latency.hip -- dependent-issue latencies on gfx950 that bound the scalar part of a MH step: a chain of
// dependent fp64 FMAs / adds, DPP move + add, LDS write -> read, workgroup barrier; with 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/dp_latency.hip -o dp_latency && ./dp_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned long long clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int MODE>
__global__ void lat(double* out, unsigned long long* ticks, double a, double b, int iters) {
  __shared__ double sh[1024];
  double x = a + threadIdx.x, y = b;
  sh[threadIdx.x] = x;
  __syncthreads();
  unsigned long long t0 = clk();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) x = __builtin_fma(x, y, a);                      // dependent fp64 fma
      if (MODE == 1) x = x + y;                                       // dependent fp64 add
      if (MODE == 2) {                                                // DPP move (2 x b32) + fp64 add
        unsigned long long ux = (unsigned long long)__double_as_longlong(x);
        unsigned lo = __builtin_amdgcn_update_dpp((unsigned)ux, (unsigned)ux, 0x128, 0xf, 0xf, true);
        unsigned hi = __builtin_amdgcn_update_dpp((unsigned)(ux >> 32), (unsigned)(ux >> 32), 0x128, 0xf, 0xf, true);
        x = x + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      }
      if (MODE == 3) {                                                // LDS write -> read (own slot)
        sh[threadIdx.x] = x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x = sh[threadIdx.x ^ 1] + y;
      }
      if (MODE == 4) {                                                // workgroup barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        x = x + y;
      }
      if (MODE == 5) {                                                // fp32 dependent fma for comparison
        float fx = (float)x; fx = __builtin_fmaf(fx, (float)y, (float)a); x = fx;
      }
      if (MODE == 6) {                                                // fp64 compare -> cndmask
        x = (x < y) ? x + a : x - a;
      }
    }
  }
  unsigned long long t1 = clk();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; unsigned long long* tk;
  hipMalloc(&out, 8 * 1024 * 512); hipMalloc(&tk, 8 * 1024);
  const char* names[] = {"fp64 fma", "fp64 add", "dpp mov x2 + fp64 add", "lds write->read + add", "s_barrier + add", "cvt+fp32 fma+cvt", "fp64 cmp+2 add+cndmask"};
  const int iters = 200;
  for (int threads : {64, 256, 512}) {
    for (int mode = 0; mode < 7; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(lat<0>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
          case 1: hipLaunchKernelGGL(lat<1>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
          case 2: hipLaunchKernelGGL(lat<2>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
          case 3: hipLaunchKernelGGL(lat<3>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
          case 4: hipLaunchKernelGGL(lat<4>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
          case 5: hipLaunchKernelGGL(lat<5>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
          default: hipLaunchKernelGGL(lat<6>, dim3(256), dim3(threads), 0, 0, out, tk, 1.0, 0.999, iters); break;
        }
        hipDeviceSynchronize();
      }
      unsigned long long h[256];
      hipMemcpy(h, tk, sizeof(h), hipMemcpyDeviceToHost);
      unsigned long long s = 0; for (auto v : h) s += v;
      printf("%3d threads/WG (%d wave(s)/SIMD): %-26s %6.1f ticks per dependent step\n", threads, threads >= 256 ? threads / 256 : 1, names[mode], (double)s / 256 / (iters * 16));
    }
  }
  return 0;
}
