// tests/cpp/small_sweep.cpp -- where does the small-matrix kernel beat the tiled kernels on device-resident operands?
// Back-to-back launches from a compiled caller (no interpreter), per shape, small path on / off.  One JSON line per shape.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "laser_hip.h"

#define CK(x)                                                         \
  do {                                                                \
    if ((x) != 0) {                                                   \
      fprintf(stderr, "FAILED %s: %s\n", #x, laser_hip_last_error()); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main() {
  CK(laser_hip_init(0));
  const int shapes[][3] = {{32, 32, 32},   {64, 64, 64},    {96, 96, 96},   {128, 128, 32},  {128, 128, 64}, {128, 128, 96},
                           {128, 128, 128}, {64, 64, 128},  {64, 128, 128}, {192, 192, 64}, {192, 192, 128}, {256, 256, 64},
                           {256, 256, 128}, {384, 384, 128}, {512, 512, 64}, {512, 512, 128}, {100, 100, 100}, {130, 130, 130}};
  float *dA, *dB, *dC;
  hipStream_t st;
  const size_t cap = 512 * 512 * sizeof(float);
  if (hipMalloc(&dA, cap) || hipMalloc(&dB, cap) || hipMalloc(&dC, cap) || hipStreamCreate(&st)) return 1;
  std::vector<float> h(512 * 512);
  unsigned s = 7;
  for (auto &v : h) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) / 16777216.0f - 0.5f) * 0.2f;
  }
  hipMemcpy(dA, h.data(), cap, hipMemcpyHostToDevice);
  hipMemcpy(dB, h.data(), cap, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (auto &sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    double us[2];
    for (int on = 1; on >= 0; on--) {
      CK(laser_hip_set_option("small_path", on));
      for (int i = 0; i < 50; i++) CK(laser_hip_gemm_strided_f32_dev(M, N, K, 1.0f, dA, K, 1, dB, N, 1, 0.0f, dC, N, 1, st));
      hipStreamSynchronize(st);
      float best = 1e30f;
      for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, st);
        for (int i = 0; i < 400; i++) CK(laser_hip_gemm_strided_f32_dev(M, N, K, 1.0f, dA, K, 1, dB, N, 1, 0.0f, dC, N, 1, st));
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      us[on] = best * 1e3 / 400;
    }
    printf("{\"shape\": [%d, %d, %d], \"small_us\": %.2f, \"tiled_us\": %.2f, \"ratio\": %.3f}\n", M, N, K, us[1], us[0], us[1] / us[0]);
  }
  laser_hip_set_option("small_path", 1);
  return 0;
}
