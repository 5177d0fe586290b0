// tests/cpp/small_gemm_bench.cpp -- BASELINE configs[0] (fp32 gemm M = N = K = 128) timed from a COMPILED caller, the
// way a Nim program would call the drop-in: no interpreter between the calls.  Prints one JSON line:
//   host_us        laser_hip_gemm_strided_f32 (host pointers, blocking), wall time per call
//   dev_us         laser_hip_gemm_strided_f32_dev (operands in HBM), back-to-back launches on one stream, per launch
//   dev_single_us  one launch from an idle stream to its completion (launch + kernel + synchronise)
//   batched_us     1000 x (32 x 32 x 32) through laser_hip_gemm_strided_batched_f32_dev, per batch launch
// and the same with the small-matrix path switched off (tiled kernels + staged copies) for comparison.
// Checks every result against a k-ordered fmaf chain computed here (= Laser's arithmetic for K <= kc = 512).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "laser_hip.h"

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x)                                                         \
  do {                                                                \
    if ((x) != 0) {                                                   \
      fprintf(stderr, "FAILED %s: %s\n", #x, laser_hip_last_error()); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

static int measure(int small_on, double *host_us, double *dev_us, double *dev_single_us, double *batched_us, int *bad) {
  const int n = 128;
  CK(laser_hip_set_option("small_path", small_on));
  std::vector<float> A(n * n), B(n * n), C(n * n), want(n * n);
  unsigned s = 12345;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) / 16777216.0f - 0.5f) * 0.2f;
  };
  for (auto &v : A) v = rnd();
  for (auto &v : B) v = rnd();
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      float acc = 0.0f;
      for (int k = 0; k < n; k++) acc = fmaf(A[i * n + k], B[k * n + j], acc);
      want[i * n + j] = acc;
    }
  // host pointers
  for (int i = 0; i < 20; i++) CK(laser_hip_gemm_strided_f32(n, n, n, 1.0f, A.data(), n, 1, B.data(), n, 1, 0.0f, C.data(), n, 1));
  const int reps = 200;
  double t0 = now_us();
  for (int i = 0; i < reps; i++) CK(laser_hip_gemm_strided_f32(n, n, n, 1.0f, A.data(), n, 1, B.data(), n, 1, 0.0f, C.data(), n, 1));
  *host_us = (now_us() - t0) / reps;
  for (int i = 0; i < n * n; i++) *bad += (C[i] != want[i]);
  // device resident
  float *dA, *dB, *dC;
  hipStream_t st;
  if (hipMalloc(&dA, n * n * 4) || hipMalloc(&dB, n * n * 4) || hipMalloc(&dC, n * n * 4) || hipStreamCreate(&st)) return 1;
  hipMemcpy(dA, A.data(), n * n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), n * n * 4, hipMemcpyHostToDevice);
  hipMemset(dC, 0xff, n * n * 4);
  for (int i = 0; i < 50; i++) CK(laser_hip_gemm_strided_f32_dev(n, n, n, 1.0f, dA, n, 1, dB, n, 1, 0.0f, dC, n, 1, st));
  hipStreamSynchronize(st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int dreps = 500;
  hipEventRecord(e0, st);
  for (int i = 0; i < dreps; i++) CK(laser_hip_gemm_strided_f32_dev(n, n, n, 1.0f, dA, n, 1, dB, n, 1, 0.0f, dC, n, 1, st));
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  *dev_us = ms * 1e3 / dreps;
  double acc_single = 0;
  for (int i = 0; i < 50; i++) {
    hipStreamSynchronize(st);
    t0 = now_us();
    CK(laser_hip_gemm_strided_f32_dev(n, n, n, 1.0f, dA, n, 1, dB, n, 1, 0.0f, dC, n, 1, st));
    hipStreamSynchronize(st);
    acc_single += now_us() - t0;
  }
  *dev_single_us = acc_single / 50;
  hipMemcpy(C.data(), dC, n * n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n * n; i++) *bad += (C[i] != want[i]);
  // batched tiny: 1000 x 32^3, operand b at ptr + b*1024
  const int nb = 1000, m = 32;
  float *bA, *bB, *bC;
  if (hipMalloc(&bA, nb * m * m * 4) || hipMalloc(&bB, nb * m * m * 4) || hipMalloc(&bC, nb * m * m * 4)) return 1;
  std::vector<float> hA(nb * m * m), hB(nb * m * m), hC(nb * m * m);
  for (auto &v : hA) v = rnd();
  for (auto &v : hB) v = rnd();
  hipMemcpy(bA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < 10; i++)
    CK(laser_hip_gemm_strided_batched_f32_dev(nb, m, m, m, 1.0f, bA, m, 1, m * m, bB, m, 1, m * m, 0.0f, bC, m, 1, m * m, st));
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < 100; i++)
    CK(laser_hip_gemm_strided_batched_f32_dev(nb, m, m, m, 1.0f, bA, m, 1, m * m, bB, m, 1, m * m, 0.0f, bC, m, 1, m * m, st));
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  hipEventElapsedTime(&ms, e0, e1);
  *batched_us = ms * 1e3 / 100;
  hipMemcpy(hC.data(), bC, hC.size() * 4, hipMemcpyDeviceToHost);
  for (int b = 0; b < nb; b += 97)
    for (int i = 0; i < m; i++)
      for (int j = 0; j < m; j++) {
        float acc = 0.0f;
        for (int k = 0; k < m; k++) acc = fmaf(hA[b * m * m + i * m + k], hB[b * m * m + k * m + j], acc);
        *bad += (hC[b * m * m + i * m + j] != acc);
      }
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(bA); hipFree(bB); hipFree(bC);
  hipStreamDestroy(st);
  return 0;
}

int main() {
  CK(laser_hip_init(0));
  double h[2], d[2], ds[2], b[2];
  int bad = 0;
  for (int on = 1; on >= 0; on--)
    if (measure(on, &h[on], &d[on], &ds[on], &b[on], &bad)) return 1;
  laser_hip_set_option("small_path", 1);
  // the host-pointer call again with the zero-copy path synchronising its stream instead of polling completion flags
  double h_sync = 0, d_tmp, ds_tmp, b_tmp;
  CK(laser_hip_set_option("zero_copy_poll", 0));
  if (measure(1, &h_sync, &d_tmp, &ds_tmp, &b_tmp, &bad)) return 1;
  CK(laser_hip_set_option("zero_copy_poll", 1));
  // Laser's own path for this size is single-threaded (M*N*K > 128^3 is false: gemm.nim:141): a plain fmaf triple loop
  // on this host for scale (the oracle's timed single-thread number is in bench_configs.py's output)
  printf("{\"config\": \"C1 fp32 gemm M=N=K=128 from a compiled caller\", \"host_us\": %.2f, \"dev_us\": %.2f, \"dev_single_us\": %.2f, "
         "\"batched_1000x32cubed_us\": %.2f, \"tiled_kernels\": {\"host_us\": %.2f, \"dev_us\": %.2f, \"dev_single_us\": %.2f, "
         "\"batched_1000x32cubed_us\": %.2f}, \"host_us_stream_synchronise\": %.2f, \"mismatches\": %d}\n",
         h[1], d[1], ds[1], b[1], h[0], d[0], ds[0], b[0], h_sync, bad);
  return bad ? 2 : 0;
}
