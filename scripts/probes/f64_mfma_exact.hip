// probe: is v_mfma_f64_16x16x4_f64 bitwise a k-ordered fma chain (like the f32 MFMA)?
// build: hipcc --offload-arch=gfx950 -O2 f64_mfma_exact.hip -o f64_mfma_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
using f64x4 = __attribute__((ext_vector_type(4))) double;
__global__ void k_mfma(const double* A, const double* B, double* C, int K) {   // A[16][K], B[K][16], C[16][16]
  const int l = threadIdx.x;
  f64x4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const double a = A[(l & 15) * K + k0 + (l >> 4)];
    const double b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main() {
  const int K = 1024;
  double *hA = (double*)malloc(16 * K * 8), *hB = (double*)malloc(16 * K * 8), hC[256], ref[256], ref2[256];
  srand(1);
  for (int i = 0; i < 16 * K; i++) { hA[i] = (rand() / (double)RAND_MAX - 0.5) * 0.2; hB[i] = (rand() / (double)RAND_MAX - 0.5) * 0.2; }
  double *dA, *dB, *dC;
  hipMalloc(&dA, 16 * K * 8); hipMalloc(&dB, 16 * K * 8); hipMalloc(&dC, 256 * 8);
  hipMemcpy(dA, hA, 16 * K * 8, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 16 * K * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  hipMemcpy(hC, dC, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    double c = 0, c2 = 0;
    for (int k = 0; k < K; k++) c = fma(hA[i * K + k], hB[k * 16 + j], c);           // ascending-k fma chain
    for (int k0 = 0; k0 < K; k0 += 4) {                                             // 4-term dot then add
      double d = 0; for (int k = k0; k < k0 + 4; k++) d = fma(hA[i * K + k], hB[k * 16 + j], d); c2 += d; }
    ref[i * 16 + j] = c; ref2[i * 16 + j] = c2;
    bad += memcmp(&c, &hC[i * 16 + j], 8) != 0; bad2 += memcmp(&c2, &hC[i * 16 + j], 8) != 0;
  }
  printf("f64 mfma vs ascending fma chain: %d/256 differ; vs per-4 partial dot: %d/256 differ; sample %.17g %.17g\n", bad, bad2, hC[5], ref[5]);
  return 0;
}
