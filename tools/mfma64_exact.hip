// Experiment: lane layout and rounding of v_mfma_f64_4x4x4_4b_f64 on gfx950.
// 1. probe with unit vectors which (A lane, B lane) pairs feed which D lane;
// 2. with that map, test whether D == sequential fp64 fma chain over the 4 products (+C first), in some order.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
__global__ void probe(unsigned char* T) {
  int l = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
      T[(la * 64 + lb) * 64 + l] = d != 0.0;
    }
}
__global__ void k(const double* A, const double* B, const double* C, double* D) {
  int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}
int main() {
  unsigned char* dT; (void)hipMalloc(&dT, 64 * 64 * 64);
  probe<<<1, 64>>>(dT);
  std::vector<unsigned char> T(64 * 64 * 64);
  (void)hipMemcpy(T.data(), dT, T.size(), hipMemcpyDeviceToHost);
  // for every D lane: list of (la, lb) contributing
  int pa[64][8], pb[64][8], np[64];
  for (int L = 0; L < 64; L++) {
    np[L] = 0;
    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++)
      if (T[(la * 64 + lb) * 64 + L] && np[L] < 8) { pa[L][np[L]] = la; pb[L][np[L]] = lb; np[L]++; }
  }
  printf("D lane <- (A lane, B lane) products:\n");
  for (int L = 0; L < 64; L += 1) {
    if (L < 8 || L % 16 == 0 || L == 63) {
      printf("  D[%2d] (%d terms):", L, np[L]);
      for (int q = 0; q < np[L]; q++) printf(" (%d,%d)", pa[L][q], pb[L][q]);
      printf("\n");
    }
  }
  double hA[64], hB[64], hC[64], hD[64], *dA, *dB, *dC, *dD;
  (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dC, 512); (void)hipMalloc(&dD, 512);
  bool ok[24]; for (auto& o : ok) o = true;
  bool ok_c_last[24]; for (auto& o : ok_c_last) o = true;
  srand(1);
  for (int trial = 0; trial < 50; trial++) {
    for (int i = 0; i < 64; i++) { hA[i] = rand() / 1e9 - 1.0; hB[i] = rand() / 1e9 - 1.0; hC[i] = rand() / 1e9 - 1.0; }
    (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD);
    (void)hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    int perm[4] = {0, 1, 2, 3};
    for (int pi = 0; pi < 24; pi++, std::next_permutation(perm, perm + 4)) {
      for (int L = 0; L < 64 && (ok[pi] || ok_c_last[pi]); L++) {
        if (np[L] != 4) { ok[pi] = ok_c_last[pi] = false; break; }
        double acc = hC[L];
        for (int q = 0; q < 4; q++) acc = fma(hA[pa[L][perm[q]]], hB[pb[L][perm[q]]], acc);
        if (acc != hD[L]) ok[pi] = false;
        double acc2 = hA[pa[L][perm[0]]] * hB[pb[L][perm[0]]];
        for (int q = 1; q < 4; q++) acc2 = fma(hA[pa[L][perm[q]]], hB[pb[L][perm[q]]], acc2);
        acc2 += hC[L];
        if (acc2 != hD[L]) ok_c_last[pi] = false;
      }
    }
  }
  int perm[4] = {0, 1, 2, 3}, n = 0;
  for (int pi = 0; pi < 24; pi++, std::next_permutation(perm, perm + 4)) {
    if (ok[pi]) { printf("BITWISE: D = fma chain starting from C, product order (by A-lane rank) %d%d%d%d\n", perm[0], perm[1], perm[2], perm[3]); n++; }
    if (ok_c_last[pi]) { printf("BITWISE: D = (fma chain of products) + C, order %d%d%d%d\n", perm[0], perm[1], perm[2], perm[3]); n++; }
  }
  if (!n) printf("no sequential fp64 fma-chain order reproduces the MFMA bitwise (wider internal accumulation?)\n");
  return 0;
}
