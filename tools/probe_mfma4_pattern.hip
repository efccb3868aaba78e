// tools/probe_mfma4_pattern.hip -- what bounds the evaluation phase of mh_sweep_mfma: cycles per (v_mfma_f64_4x4x4_4b +
// dependent fma) pair in the kernel's own issue pattern (batches of MB independent MFMAs on 80 distinct A registers, then
// their MB fma(d, d, acc)), at one and two waves per SIMD, and with UNEVEN shares of the two waves of a SIMD (waves 0..3
// run NA pairs, waves 4..7 NB pairs, one LDS barrier per iteration) -- the shape a speculative head start of the
// partner waves would leave behind the decision.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_mfma4_pattern.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int MB, int NPAIR, bool FMA, bool MFMA>
__device__ __forceinline__ void body(const double (&a)[80], double b, double c, double (&acc)[4]) {
#pragma unroll
  for (int t0 = 0; t0 < NPAIR; t0 += MB) {
    double d[MB];
#pragma unroll
    for (int u = 0; u < MB; u++)
      if (t0 + u < NPAIR) d[u] = MFMA ? __builtin_amdgcn_mfma_f64_4x4x4f64(a[t0 + u], b, c, 0, 0, 0) : a[t0 + u];
#pragma unroll
    for (int u = 0; u < MB; u++)
      if (t0 + u < NPAIR) {
        if (FMA) acc[u & 3] = __builtin_fma(d[u], d[u], acc[u & 3]);
        else asm volatile("" : : "v"(d[u]));   // (keeps the MFMA alive, no instruction)
      }
  }
}

// MODE 0: both waves NA pairs (the kernel today)   1: waves 4..7 NB pairs   2: MFMA only   3: fma only
template <int MB, int NA, int NB, int MODE>
__global__ __launch_bounds__(512) void k(const double* in, double* out, unsigned long long* ticks, int iters) {
  double a[80];
#pragma unroll
  for (int t = 0; t < 80; t++) a[t] = in[t * 512 + threadIdx.x];
  double acc[4] = {0, 0, 0, 0};
  double b = in[threadIdx.x & 63], c = 1e-3;
  const bool late = (threadIdx.x >> 6) >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    b = b + 1e-9;
    if (MODE == 1 && late) body<MB, NB, true, true>(a, b, c, acc);
    else if (MODE == 2) body<MB, NA, false, true>(a, b, c, acc);
    else if (MODE == 3) body<MB, NA, true, false>(a, b, c, acc);
    else body<MB, NA, true, true>(a, b, c, acc);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MB, int NA, int NB, int MODE>
void run(const char* name, int bs, const double* in, double* out, unsigned long long* ticks) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MB, NA, NB, MODE><<<blocks, bs>>>(in, out, ticks, 100); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<MB, NA, NB, MODE><<<blocks, bs>>>(in, out, ticks, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[blocks / 2] / iters;
  const int pairs_simd = (bs == 512) ? (MODE == 1 ? NA + NB : 2 * NA) : NA;
  printf("%-34s block=%3d MB=%2d: %7.1f cycles/iter (s_memtime), %6.2f cycles per pair and SIMD, wall %.1f ns/iter\n",
         name, bs, MB, med, pairs_simd ? med / pairs_simd : 0.0, ms * 1e6 / iters);
}

int main() {
  double *in, *out; unsigned long long* ticks;
  (void)hipMalloc(&in, 80 * 512 * 8); (void)hipMalloc(&out, 256 * 512 * 8); (void)hipMalloc(&ticks, 256 * 8);
  std::vector<double> h(80 * 512);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
  (void)hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  for (int bs : {256, 512}) {
    if (bs == 256) {
      run<8, 80, 80, 0>("80 pairs per wave", 256, in, out, ticks);
      run<8, 80, 80, 2>("80 MFMA only", 256, in, out, ticks);
      run<8, 80, 80, 3>("80 fma only", 256, in, out, ticks);
      run<4, 80, 80, 0>("80 pairs per wave", 256, in, out, ticks);
      run<16, 80, 80, 0>("80 pairs per wave", 256, in, out, ticks);
    } else {
      run<8, 80, 80, 0>("80 + 80 pairs", 512, in, out, ticks);
      run<8, 80, 80, 2>("80 + 80 MFMA only", 512, in, out, ticks);
      run<8, 80, 80, 3>("80 + 80 fma only", 512, in, out, ticks);
      run<4, 80, 80, 0>("80 + 80 pairs", 512, in, out, ticks);
      run<16, 80, 80, 0>("80 + 80 pairs", 512, in, out, ticks);
      run<8, 80, 40, 1>("80 + 40 pairs (head start right)", 512, in, out, ticks);
      run<8, 80, 24, 1>("80 + 24 pairs", 512, in, out, ticks);
      run<8, 80, 0, 1>("80 + 0 pairs", 512, in, out, ticks);
      run<8, 40, 40, 0>("40 + 40 pairs", 512, in, out, ticks);
    }
  }
  return 0;
}
