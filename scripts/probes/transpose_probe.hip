// scripts/probes/transpose_probe.hip -- A/B of 2-D transpose kernels at the reference's bench shape (4000 x 2000 f32,
// benchmarks/transpose/transpose_bench.nim:521-527) and larger ones: the library's kernel (through the C-ABI) against
// candidates defined here, timed from C++ (hipEvents over back-to-back launches) so that no interpreter sits between launches.
// build: make -C scripts/probes transpose_probe   run: LD_LIBRARY_PATH=laser_amd/lib scripts/probes/transpose_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "../../laser_amd/csrc/data_movement.hip"   // the library's kernel template, for tile-shape variants
extern "C" int laser_hip_init(int);
extern "C" int laser_hip_transpose2d_batched_b32_dev(void *dst, const void *src, int64_t n, int64_t rows, int64_t cols, void *stream);

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// candidate R: no LDS -- lane (i, j) of an 8 x 8 lane grid owns a 4 x 4 block: 4 loads of 16 B (8 rows x 128-B lines per
// instruction), the 4 x 4 transpose is register renaming, 4 stores of 16 B (8 destination rows x 128-B lines per instruction).
// One wave = a 32 x 32 tile; TPW tiles per wave in flight.
template <int TPW>
__global__ void __launch_bounds__(256) transpose_reg_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int64_t NR, int64_t NC,
                                                            int64_t tiles_c, int64_t tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane >> 3, j = lane & 7;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * TPW;
  u32x4 q[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    const int64_t tile = t0 + t;
    const int64_t r = (tile / tiles_c) * 32 + 4 * i, c = (tile % tiles_c) * 32 + 4 * j;
#pragma unroll
    for (int k = 0; k < 4; k++)
      q[t][k] = (tile < tiles && r + k < NR && c < NC) ? *reinterpret_cast<const u32x4 *>(src + (r + k) * NC + c) : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    const int64_t tile = t0 + t;
    const int64_t r = (tile / tiles_c) * 32 + 4 * i, c = (tile % tiles_c) * 32 + 4 * j;
    if (tile < tiles && r < NR) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (c + k < NC) *reinterpret_cast<u32x4 *>(dst + (c + k) * NR + r) = u32x4{q[t][0][k], q[t][1][k], q[t][2][k], q[t][3][k]};
    }
  }
}

// plain copy of the same bytes (the yardstick)
__global__ void __launch_bounds__(256) copy_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (i + 256 * k < n) dst[i + 256 * k] = src[i + 256 * k];
}

template <typename F>
static double time_us(F fn, int inner = 50) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 200; w++) fn();
  hipDeviceSynchronize();
  std::vector<double> ts;
  for (int rep = 0; rep < 7; rep++) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < inner; k++) fn();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1000.0 / inner);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  laser_hip_init(0);
  const int64_t shapes[][2] = {{4000, 2000}, {4096, 4096}, {8192, 8192}, {16384, 8192}, {2000, 4000}};
  for (auto &sh : shapes) {
    const int64_t NR = sh[0], NC = sh[1], n = NR * NC;
    uint32_t *s, *d;
    hipMalloc((void **)&s, n * 4); hipMalloc((void **)&d, n * 4);
    std::vector<uint32_t> h(n);
    for (int64_t k = 0; k < n; k++) h[k] = (uint32_t)(k * 2654435761u);
    hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
    const double bytes = 2.0 * n * 4;
    const int64_t tiles_r = (NR + 31) / 32, tiles_c = (NC + 31) / 32, tiles = tiles_r * tiles_c;
    auto check = [&](const char *name) {
      std::vector<uint32_t> o(n);
      hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
      int64_t bad = 0;
      for (int64_t r = 0; r < NR && bad == 0; r += 7)
        for (int64_t c = 0; c < NC; c += 3) bad += o[c * NR + r] != h[r * NC + c];
      if (bad) printf("  %s: WRONG\n", name);
    };
    const double t_lib = time_us([&] { laser_hip_transpose2d_batched_b32_dev(d, s, 1, NR, NC, nullptr); });
    check("lib");
    hipMemset(d, 0, n * 4);
    const double t_r1 = time_us([&] { hipLaunchKernelGGL(transpose_reg_kernel<1>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, d, s, NR, NC, tiles_c, tiles); });
    check("reg1");
    hipMemset(d, 0, n * 4);
    const double t_r2 = time_us([&] { hipLaunchKernelGGL(transpose_reg_kernel<2>, dim3((unsigned)((tiles + 7) / 8)), dim3(256), 0, 0, d, s, NR, NC, tiles_c, tiles); });
    check("reg2");
    hipMemset(d, 0, n * 4);
    const double t_r4 = time_us([&] { hipLaunchKernelGGL(transpose_reg_kernel<4>, dim3((unsigned)((tiles + 15) / 16)), dim3(256), 0, 0, d, s, NR, NC, tiles_c, tiles); });
    check("reg4");
    auto lds = [&](auto trc, auto tcc) {
      constexpr int TR = decltype(trc)::value, TC = decltype(tcc)::value;
      const int64_t tr_ = (NR + TR - 1) / TR, tc_ = (NC + TC - 1) / TC;
      hipMemset(d, 0, n * 4);
      const double t = time_us([&] { hipLaunchKernelGGL((laser_hip::transpose_batched_kernel<uint32_t, true, TR, TC, false>), dim3((unsigned)(tr_ * tc_)), dim3(256), 0, 0, d, s, NR, NC, tc_, tr_); });
      check("lds");
      printf("  lds %dx%d: %.2f us %.0f GB/s\n", TR, TC, t, bytes / t / 1e3);
    };
    lds(std::integral_constant<int, 64>{}, std::integral_constant<int, 128>{});
    lds(std::integral_constant<int, 32>{}, std::integral_constant<int, 128>{});
    lds(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
    lds(std::integral_constant<int, 32>{}, std::integral_constant<int, 256>{});
    lds(std::integral_constant<int, 128>{}, std::integral_constant<int, 64>{});
    lds(std::integral_constant<int, 16>{}, std::integral_constant<int, 256>{});
    const double t_cp = time_us([&] { hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (u32x4 *)d, (const u32x4 *)s, n / 4); });
    const double t_mc = time_us([&] { hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, 0); });
    printf("{\"shape\": [%ld, %ld], \"lib_us\": %.2f, \"lib_gbps\": %.0f, \"reg1_us\": %.2f, \"reg1_gbps\": %.0f, \"reg2_us\": %.2f, \"reg2_gbps\": %.0f, \"reg4_us\": %.2f, \"reg4_gbps\": %.0f, "
           "\"copy_kernel_us\": %.2f, \"copy_kernel_gbps\": %.0f, \"memcpy_d2d_us\": %.2f, \"memcpy_d2d_gbps\": %.0f}\n",
           (long)NR, (long)NC, t_lib, bytes / t_lib / 1e3, t_r1, bytes / t_r1 / 1e3, t_r2, bytes / t_r2 / 1e3, t_r4, bytes / t_r4 / 1e3, t_cp, bytes / t_cp / 1e3, t_mc, bytes / t_mc / 1e3);
    fflush(stdout);
    hipFree(s); hipFree(d);
  }
  return 0;
}
