// tools/grid_barrier_custom.hip -- a two-level grid barrier (8 group counters + 1 top counter, monotonically increasing,
// one lane per workgroup) against cooperative_groups::grid_group::sync (27 us at 256 workgroups, tools/
// grid_sync_latency.hip): what a per-step hand-over between ALL workgroups would cost with a hand-written barrier.
// Every barrier also hands one double per workgroup to its neighbour and checks it (visibility across XCDs).
// Launched cooperatively (all workgroups co-resident) and with a bounded spin (gives up and reports instead of hanging).
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef BAR_VARIANT
#define BAR_VARIANT 1
#endif
// grp: 8 arrival counters (128 B apart), top: arrival counter of the groups, rel: 8 release words (128 B apart, variant 1)
__device__ __forceinline__ bool grid_barrier(unsigned* grp, unsigned* top, unsigned epoch, unsigned ngroups, unsigned gsize) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x % ngroups;
    unsigned* rel = top + 32;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(&grp[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * gsize) {
      const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (BAR_VARIANT == 1 && t + 1 == epoch * ngroups)       // last group in: release everybody through 8 separate lines
        for (unsigned q = 0; q < ngroups; q++) __hip_atomic_store(&rel[q * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    if (BAR_VARIANT == 1) {
      while (__hip_atomic_load(&rel[g * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 8000000u) { ok = false; break; }
      }
    } else {
      while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * ngroups) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 8000000u) { ok = false; break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}
__global__ __launch_bounds__(512) void k(double* buf, unsigned* grp, unsigned* top, int iters, int* err) {
  const unsigned ngroups = 8, gsize = gridDim.x / 8;
  double v = 0.0;
  for (int i = 1; i <= iters; i++) {
    if (threadIdx.x == 0) __hip_atomic_store(&buf[blockIdx.x], (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_barrier(grp, top, (unsigned)(2 * i - 1), ngroups, gsize)) { if (threadIdx.x == 0) atomicAdd(err, 1000000); return; }
    const double got = __hip_atomic_load(&buf[(blockIdx.x + 37) % gridDim.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && got != (double)i) atomicAdd(err, 1);
    v += got;
    if (!grid_barrier(grp, top, (unsigned)(2 * i), ngroups, gsize)) { if (threadIdx.x == 0) atomicAdd(err, 1000000); return; }   // (buf is rewritten next)
  }
  if (threadIdx.x == 0) buf[gridDim.x + blockIdx.x] = v;
}
int main() {
  double* buf; unsigned *grp, *top; int* err;
  hipMalloc(&buf, 8 * 4096); hipMalloc(&grp, 4 * 32 * 8); hipMalloc(&top, 128 * 10); hipMalloc(&err, 4);
  for (int blocks : {64, 128, 256}) {
    for (int iters : {200, 2000}) {
      hipMemset(buf, 0, 8 * 4096); hipMemset(grp, 0, 4 * 32 * 8); hipMemset(top, 0, 128 * 10); hipMemset(err, 0, 4);
      void* args[] = {&buf, &grp, &top, &iters, &err};
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipError_t e = hipLaunchCooperativeKernel((const void*)k, dim3(blocks), dim3(512), args, 0, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int herr = -1; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("blocks %3d iters %5d: %s, %.3f ms -> %.2f us per barrier (2 per iteration), errors %d\n", blocks, iters, hipGetErrorString(e), ms, ms * 1e3 / (2 * iters), herr);
    }
  }
  return 0;
}
