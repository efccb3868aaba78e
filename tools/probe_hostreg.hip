// tools/probe_hostreg.hip -- what it costs to pin a caller's (pageable) result array for the duration of a call, and what the
// copies into it then run at: hipHostRegister / hipHostUnregister of 410 MB, 1-D and 2-D (row-range) device-to-host copies into
// pageable, registered and hipHostMalloc'ed memory.   hipcc --offload-arch=gfx950 -O2 tools/probe_hostreg.hip -o probe_hostreg
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  const size_t rows = 5120, S = 10000, bytes = rows * S * 8;      // [chain x column][kept row]: 410 MB
  double* dev; CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 1, bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  double* page = (double*)malloc(bytes); memset(page, 0, bytes);
  double* pinned; CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault)); memset(pinned, 0, bytes);
  auto copy1d = [&](double* dst, const char* what) {
    for (int r = 0; r < 2; r++) {
      const double t0 = now();
      hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, st); const double t1 = now(); hipStreamSynchronize(st);
      const double t2 = now();
      if (r) printf("%-34s 1-D 410 MB: call returns after %6.2f ms, done after %6.2f ms = %5.1f GB/s\n", what, t1 - t0, t2 - t0, bytes / (t2 - t0) * 1e-6);
    }
  };
  auto copy2d = [&](double* dst, const char* what) {
    for (int r = 0; r < 2; r++) {
      const double t0 = now();
      for (int q = 0; q < 8; q++)
        hipMemcpy2DAsync(dst + q * (S / 8), S * 8, dev + q * (S / 8), S * 8, (S / 8) * 8, rows, hipMemcpyDeviceToHost, st);
      const double t1 = now(); hipStreamSynchronize(st);
      const double t2 = now();
      if (r) printf("%-34s 2-D 8 x 51 MB: calls return after %6.2f ms, done after %6.2f ms = %5.1f GB/s\n", what, t1 - t0, t2 - t0, bytes / (t2 - t0) * 1e-6);
    }
  };
  copy1d(page, "pageable"); copy2d(page, "pageable");
  copy1d(pinned, "hipHostMalloc"); copy2d(pinned, "hipHostMalloc");
  for (int r = 0; r < 3; r++) {
    const double t0 = now();
    CK(hipHostRegister(page, bytes, hipHostRegisterDefault));
    const double t1 = now();
    if (r == 2) { copy1d(page, "registered"); copy2d(page, "registered"); }
    const double t2 = now();
    CK(hipHostUnregister(page));
    const double t3 = now();
    printf("hipHostRegister(410 MB) %6.2f ms, hipHostUnregister %6.2f ms\n", t1 - t0, t3 - t2);
  }
  // registering in pieces (what a call would do under its first sub-call)
  {
    const double t0 = now();
    for (int q = 0; q < 8; q++) CK(hipHostRegister((char*)page + q * (bytes / 8), bytes / 8, hipHostRegisterDefault));
    const double t1 = now();
    for (int q = 0; q < 8; q++) CK(hipHostUnregister((char*)page + q * (bytes / 8)));
    printf("8 x hipHostRegister(51 MB) %6.2f ms, unregister %6.2f ms\n", t1 - t0, now() - t1);
  }
  return 0;
}
