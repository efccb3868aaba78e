// Probe: the K-block of shard_columns_mfma's T10 form -- two v_mfma_f64_16x16x4 (independent accumulators) and two
// v_mfma_f64_4x4x4_4b (two more chains) -- register operands only, ONE wave per SIMD (block 256) or two (block 512).
// Ideal: 2 x 64 + 2 x 16 = 160 cycles per block.  Variants: order of the four, and 4 chains of 4x4x4 instead of 2.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/probe_mfma_mix.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ void k(double* out, int iters, double a, double b) {
  d4 c0 = {0, 1, 2, 3}, c1 = {4, 5, 6, 7}, c2 = {1, 1, 1, 1};
  double e0 = 1, e1 = 2, e2 = 3, e3 = 4;
  double x = a + threadIdx.x * 1e-6, y = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int kb = 0; kb < 12; kb++) {
      if (V == 0) {          // 16, 16, 4, 4
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
        e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e1, 0, 0, 0);
      } else if (V == 1) {   // 16, 4, 16, 4
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e1, 0, 0, 0);
      } else if (V == 2) {   // three 16x16x4 (the form before T10)
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
      } else if (V == 3) {   // only the two 16x16x4
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
      } else if (V == 4) {   // only the two 4x4x4 chains
        e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e1, 0, 0, 0);
      } else if (V == 5) {   // four 4x4x4 chains (two K-blocks' worth: even / odd columns split -- NOT the canonical chain; timing only)
        e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e1, 0, 0, 0);
        e2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e2, 0, 0, 0);
        e3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e3, 0, 0, 0);
      } else if (V == 6) {   // one 16x16x4 + the 4x4x4 pair
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, e1, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = c0[0] + c0[3] + c1[0] + c1[2] + c2[1] + e0 + e1 + e2 + e3;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / ((double)iters * 12);
  if (s == 12345.678) out[1] = s;
}
template <int V> void run(double* out, int bs, const char* what) {
  k<V><<<256, bs>>>(out, 4000, 1.0000001, 1e-9); (void)hipDeviceSynchronize();
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a); k<V><<<256, bs>>>(out, 4000, 1.0000001, 1e-9); (void)hipEventRecord(b); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("block %4d  %-34s %.1f s_memtime ticks per K-block, wall %.1f ns per K-block\n", bs, what, h, ms * 1e6 / (4000.0 * 12));
}
int main() {
  double* out; (void)hipMalloc(&out, 64);
  for (int bs : {256, 512}) {
    run<0>(out, bs, "16 16 4 4"); run<1>(out, bs, "16 4 16 4"); run<2>(out, bs, "16 16 16"); run<3>(out, bs, "16 16");
    run<4>(out, bs, "4 4 (two chains)"); run<5>(out, bs, "4 4 4 4 (four chains)"); run<6>(out, bs, "16 4 4");
  }
  return 0;
}
