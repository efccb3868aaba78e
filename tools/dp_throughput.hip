// tools/dp_throughput.hip -- what INDEPENDENT fp64 FMAs actually sustain on gfx950, per SIMD, with 1..4 waves on it:
// 20 independent accumulators per thread, operands all-VGPR or one SGPR (the shape of eval_sharded's column loop).
// dp_latency.hip says one wave issues a dependent or independent VALU op every ~7 cycles; this asks whether a second /
// third / fourth wave on the SIMD fills the gaps up to the nominal 4 cycles per wave-instruction (78.6 TFLOP/s).
//   hipcc --offload-arch=gfx950 -O3 tools/dp_throughput.hip -o dp_throughput && ./dp_throughput
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void thr(double* out, const double* in, int iters) {
  double acc[20];
#pragma unroll
  for (int i = 0; i < 20; i++) acc[i] = in[i] + threadIdx.x;
  double y = in[20 + (threadIdx.x & 1)];
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int i = 0; i < 20; i++) acc[i] = __builtin_fma(acc[i], y, y);
    } else {
      // one SGPR multiplicand per FMA, 20 scalars loaded per group like a half column
      const double __attribute__((address_space(4)))* sp =
          (const double __attribute__((address_space(4)))*)(unsigned long long)(in + 32 + (it & 7) * 80);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int i = 0; i < 20; i++) acc[i] = __builtin_fma(sp[u * 20 + i], y, acc[i]);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 20; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *out, *in;
  hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&in, 8 * 1024);
  double h[1024]; for (int i = 0; i < 1024; i++) h[i] = 1.0 / (1 + i);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 2; mode++)
    for (int threads : {256, 512, 768, 1024}) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(thr<0>, dim3(256), dim3(threads), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(thr<1>, dim3(256), dim3(threads), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double fma_wave_instr_per_simd = (double)iters * 80 * (threads / 64) / 4.0;
      printf("%s, %d wave(s) per SIMD: %.1f TFLOP/s, %.2f ns per wave-FMA per SIMD (= %.2f cycles at 2.4 GHz)\n",
             mode ? "SGPR x VGPR + VGPR" : "all-VGPR", threads / 256,
             2.0 * iters * 80 * 256.0 * threads / (ms * 1e-3) / 1e12, ms * 1e6 / fma_wave_instr_per_simd,
             ms * 1e6 / fma_wave_instr_per_simd * 2.4);
    }
  return 0;
}
