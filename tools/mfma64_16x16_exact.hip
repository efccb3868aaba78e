// Experiment: lane layout and rounding of v_mfma_f64_16x16x4_f64 on gfx950 (the tile a bit-exact fp64 GEMM -- e.g. the
// observation-sharded evaluation of wide models, DESIGN.md C4 -- would be built from).
// 1. probe with unit vectors which (A lane, B lane) pairs feed which (D lane, D register);
// 2. with that map, test whether D == sequential fp64 fma chain over the 4 products starting from C, in some order.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned char* T) {
  int l = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      d4 c = {0.0, 0.0, 0.0, 0.0};
      d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, c, 0, 0, 0);
      for (int r = 0; r < 4; r++) T[((la * 64 + lb) * 64 + l) * 4 + r] = d[r] != 0.0;
    }
}
__global__ void k(const double* A, const double* B, const double* C, double* D) {
  int l = threadIdx.x;
  d4 c = {C[l * 4], C[l * 4 + 1], C[l * 4 + 2], C[l * 4 + 3]};
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = d[r];
}
int main() {
  unsigned char* dT; (void)hipMalloc(&dT, 64 * 64 * 64 * 4);
  probe<<<1, 64>>>(dT);
  std::vector<unsigned char> T(64 * 64 * 64 * 4);
  (void)hipMemcpy(T.data(), dT, T.size(), hipMemcpyDeviceToHost);
  static int pa[256][8], pb[256][8], np[256];
  for (int o = 0; o < 256; o++) {          // o = D lane * 4 + register
    np[o] = 0;
    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++)
      if (T[((la * 64 + lb) * 64 + o / 4) * 4 + o % 4] && np[o] < 8) { pa[o][np[o]] = la; pb[o][np[o]] = lb; np[o]++; }
  }
  printf("D (lane, reg) <- (A lane, B lane) products:\n");
  for (int o : {0, 1, 2, 3, 4, 5, 64, 65, 128, 255}) {
    printf("  D[%2d].%d (%d terms):", o / 4, o % 4, np[o]);
    for (int q = 0; q < np[o]; q++) printf(" (%d,%d)", pa[o][q], pb[o][q]);
    printf("\n");
  }
  double hA[64], hB[64], hC[256], hD[256], *dA, *dB, *dC, *dD;
  (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dC, 2048); (void)hipMalloc(&dD, 2048);
  bool ok[24]; for (auto& x : ok) x = true;
  srand(1);
  for (int trial = 0; trial < 50; trial++) {
    for (int i = 0; i < 64; i++) { hA[i] = rand() / 1e9 - 1.0; hB[i] = rand() / 1e9 - 1.0; }
    for (int i = 0; i < 256; i++) hC[i] = rand() / 1e9 - 1.0;
    (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD);
    (void)hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    int perm[4] = {0, 1, 2, 3};
    for (int pi = 0; pi < 24; pi++, std::next_permutation(perm, perm + 4)) {
      for (int o = 0; o < 256 && ok[pi]; o++) {
        if (np[o] != 4) { ok[pi] = false; break; }
        double acc = hC[o];
        for (int q = 0; q < 4; q++) acc = fma(hA[pa[o][perm[q]]], hB[pb[o][perm[q]]], acc);
        if (acc != hD[o]) ok[pi] = false;
      }
    }
  }
  int perm[4] = {0, 1, 2, 3}, n = 0;
  for (int pi = 0; pi < 24; pi++, std::next_permutation(perm, perm + 4))
    if (ok[pi]) { printf("BITWISE: D = fma chain starting from C, product order (by A-lane rank) %d%d%d%d\n", perm[0], perm[1], perm[2], perm[3]); n++; }
  if (!n) printf("no sequential fp64 fma-chain order reproduces v_mfma_f64_16x16x4 bitwise\n");
  return 0;
}
