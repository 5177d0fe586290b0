// Probe build of the conv pixel-tail kernel (laser_amd/csrc/conv_tail.hip compiled with CT_TIMING): the kernel alone on C4's tail
// (32 images x 64 pixels x 256 channels, K = 1152), back-to-back launch time and s_memtime stamps of its phases per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I laser_amd/csrc scripts/probes/conv_tail_timing.hip -o /tmp/ct_probe && /tmp/ct_probe
#define CT_TIMING 1
#include "../../laser_amd/csrc/conv_tail.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>


#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  const int batch = 32, Cin = 128, H = 56, W = 56, M = 256, K = Cin * 9, npix = H * W, cut = argc > 1 ? atoi(argv[1]) : 3072;
  std::vector<float> hf((size_t)M * K), hi((size_t)batch * Cin * H * W);
  srand(7);
  for (auto &v : hf) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto &v : hi) v = (float)rand() / RAND_MAX - 0.5f;
  float *df, *di, *dout; unsigned long long *dbg;
  CK(hipMalloc(&df, hf.size() * 4)); CK(hipMalloc(&di, hi.size() * 4)); CK(hipMalloc(&dout, (size_t)batch * M * npix * 4));
  CK(hipMalloc(&dbg, (size_t)batch * 8 * 8 * 8 * 8)); CK(hipMemset(dbg, 0, (size_t)batch * 8 * 8 * 8 * 8));
  CK(hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0, (size_t)batch * M * npix * 4));
  laser_hip::g_ct_dbg = dbg;
  laser_hip::GemmArgs<float> a{};
  a.M = M; a.N = npix; a.K = K; a.alpha = 1.0f; a.beta = 0.0f;
  a.A = df; a.rsA = K; a.csA = 1; a.B = di; a.bsB = (int64_t)Cin * H * W; a.C = dout; a.rsC = npix; a.csC = 1; a.bsC = (int64_t)M * npix;
  a.batch = batch; a.cH = H; a.cW = W; a.ckH = 3; a.ckW = 3; a.coW = W; a.cpH = 1; a.cpW = 1; a.csH = 1; a.csW = 1; a.col0 = cut;
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int i = 0; i < 30; i++) CK(laser_hip::launch_conv_tail_f32(a, 512, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 300;
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < n; i++) CK(laser_hip::launch_conv_tail_f32(a, 512, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"back_to_back_us_per_launch\": %.2f", ms * 1000.0f / n);
  const int mblks = M / 32, nwg = batch * mblks;
  std::vector<unsigned long long> h((size_t)nwg * 64);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  // per wave class (wave index within the workgroup): mean of stamp[i] - stamp[0] over the workgroups
  const char *names[7] = {"start", "table+barrier", "prologue_done", "first_task_done", "all_tasks_done", "after_barrier", "end"};
  unsigned long long kmin = ~0ull, kmax = 0;
  for (int w = 0; w < 8; w++) {
    double sum[7] = {0}; int cnt = 0;
    for (int g = 0; g < nwg; g++) {
      const unsigned long long *st = &h[((size_t)g * 8 + w) * 8];
      if (st[0] == 0 || st[6] == 0) continue;
      cnt++;
      for (int i = 0; i < 7; i++) sum[i] += (double)(st[i] - st[0]);
      if (st[0] < kmin) kmin = st[0];
      if (st[6] > kmax) kmax = st[6];
    }
    if (!cnt) continue;
    printf(", \"wave%d\": {", w);
    for (int i = 1; i < 7; i++) printf("%s\"%s\": %.0f", i > 1 ? ", " : "", names[i], sum[i] / cnt);
    printf("}");
  }
  printf(", \"first_start_to_last_end_ticks\": %llu}\n", kmax - kmin);
  return 0;
}
