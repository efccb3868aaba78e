// tools/probe_dp_ops.hip -- issue cost of the fp64-side instructions a range reduction can be built from (gfx950): cycles per
// wave-instruction per SIMD with 2 waves per SIMD, 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dp_ops.hip -o probe_dp_ops && ./probe_dp_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OP1(name, insn)                                                                           \
  __global__ void k_##name(double* out, const double* in, int iters) {                            \
    double a[8];                                                                                  \
    for (int i = 0; i < 8; i++) a[i] = in[i] + threadIdx.x * 1e-3;                                \
    for (int it = 0; it < iters; it++) {                                                          \
      _Pragma("unroll") for (int u = 0; u < 4; u++)                                               \
      _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(insn : "+v"(a[i]));              \
    }                                                                                             \
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
  }
OP1(fma, "v_fma_f64 %0, %0, %0, %0")
OP1(add, "v_add_f64 %0, %0, %0")
OP1(mul, "v_mul_f64 %0, %0, %0")
OP1(fract, "v_fract_f64 %0, %0")
OP1(floor, "v_floor_f64 %0, %0")
OP1(rndne, "v_rndne_f64 %0, %0")
OP1(trunc, "v_trunc_f64 %0, %0")
OP1(min, "v_min_f64 %0, %0, %0")
OP1(mov, "v_mov_b64 %0, %0")
__global__ void k_cvtu(double* out, const double* in, int iters) {
  double a[8]; unsigned r[8];
  for (int i = 0; i < 8; i++) { a[i] = in[i] + threadIdx.x * 1e-3; r[i] = 0; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));
  }
  double s = 0; for (int i = 0; i < 8; i++) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cvti(double* out, const double* in, int iters) {
  double a[8]; int r[8];
  for (int i = 0; i < 8; i++) { a[i] = in[i] + threadIdx.x * 1e-3; r[i] = 0; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));
  }
  double s = 0; for (int i = 0; i < 8; i++) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshl(double* out, const double* in, int iters) {
  unsigned r[8];
  for (int i = 0; i < 8; i++) r[i] = (unsigned)in[i] + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_lshl_add_u32 %0, %0, 4, %0" : "+v"(r[i]));
  }
  double s = 0; for (int i = 0; i < 8; i++) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// mixes (round 4, second series): does a scalar / LDS instruction still cost vector issue time when MORE waves share the SIMD?
// per 32 fma: NS scalar adds + NL ds_read_b128 (independent of the fma chains); cycles are per FMA.
template <int NS, int NL>
__global__ void __launch_bounds__(1024) k_mixt(double* out, const double* in, int iters) {
  __shared__ double sm[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = in[i & 63];
  __syncthreads();
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = in[i] + threadIdx.x * 1e-3;
  unsigned sacc = 0, addr = (threadIdx.x & 63) * 16;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 l[4] = {};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
        if ((u * 8 + i) % (32 / (NS ? NS : 1)) == 0 && NS) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if ((u * 8 + i) % (32 / (NL ? NL : 1)) == 0 && NL) asm volatile("ds_read_b128 %0, %1" : "=v"(l[((u * 8 + i) / (32 / (NL ? NL : 1))) & 3]) : "v"(addr));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  double s = sacc; for (int i = 0; i < 8; i++) s += a[i];
  for (int i = 0; i < 4; i++) s += l[i][0] + l[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *out, *in;
  hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&in, 8 * 64);
  double h[64]; for (int i = 0; i < 64; i++) h[i] = 1.0 + 1.0 / (1 + i);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
#define RUN(name)                                                                                                   \
  for (int threads : {256, 512, 768, 1024}) {                                                                                   \
    float ms = 0;                                                                                                    \
    for (int rep = 0; rep < 2; rep++) {                                                                              \
      hipEventRecord(e0); hipLaunchKernelGGL(k_##name, dim3(256), dim3(threads), 0, 0, out, in, iters);              \
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                                 \
    }                                                                                                                \
    const double per_simd = (double)iters * 32 * (threads / 256);                                                    \
    printf("%-6s %d wave(s)/SIMD: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", #name, threads / 256,   \
           ms * 1e-3 * 2.4e9 / per_simd);                                                                            \
  }
  RUN(fma) RUN(add) RUN(mul) RUN(fract) RUN(floor) RUN(rndne) RUN(trunc) RUN(min) RUN(mov) RUN(cvtu) RUN(cvti) RUN(lshl)
#define RUNMIX(NS_, NL_)                                                                                             \
  for (int threads : {256, 512, 768, 1024}) {                                                                        \
    float ms = 0;                                                                                                    \
    for (int rep = 0; rep < 2; rep++) {                                                                              \
      hipEventRecord(e0); hipLaunchKernelGGL((k_mixt<NS_, NL_>), dim3(256), dim3(threads), 0, 0, out, in, iters);    \
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                                 \
    }                                                                                                                \
    const double per_simd = (double)iters * 32 * (threads / 256);                                                    \
    printf("mix 32 fma + %d salu + %d ds_read_b128, %d wave(s)/SIMD: %.2f cycles per FMA per SIMD (at 2.4 GHz)\n", NS_, NL_, \
           threads / 256, ms * 1e-3 * 2.4e9 / per_simd);                                                             \
  }
  RUNMIX(0, 0) RUNMIX(8, 0) RUNMIX(0, 8) RUNMIX(8, 8) RUNMIX(16, 8)
  return 0;
}
