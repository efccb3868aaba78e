// tools/grid_sync_latency.hip -- cost of a grid-wide barrier (cooperative launch, cooperative_groups::grid_group::sync)
// with one 512-thread workgroup per CU: the price of handing data between ALL workgroups once per MH step.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_sync_latency.hip -o grid_sync_latency
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(512) void k(double* buf, int iters) {
  cg::grid_group g = cg::this_grid();
  double v = threadIdx.x;
  for (int i = 0; i < iters; i++) {
    if (threadIdx.x == 0) buf[blockIdx.x] = v + i;        // a small hand-over per barrier, read back from the next block
    g.sync();
    v += buf[(blockIdx.x + 1) % gridDim.x];
  }
  if (threadIdx.x == 0) buf[gridDim.x + blockIdx.x] = v;
}
int main() {
  int dev = 0, ncu = 0, coop = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
  printf("CUs %d, cooperative launch %d\n", ncu, coop);
  double* buf; hipMalloc(&buf, 8 * 4096); hipMemset(buf, 0, 8 * 4096);
  for (int blocks : {ncu / 4, ncu / 2, ncu}) {
    for (int iters : {100, 2000}) {
      void* args[] = {&buf, &iters};
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipError_t e = hipLaunchCooperativeKernel((const void*)k, dim3(blocks), dim3(512), args, 0, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %3d iters %5d: %s, %.3f ms -> %.2f us per barrier (incl. launch)\n", blocks, iters, hipGetErrorString(e), ms, ms * 1e3 / iters);
    }
  }
  return 0;
}
