// What does gfx950 return for raw buffer loads that are unaligned, negative (wrapped) or straddle
// num_records?  Descriptor sits in the middle of a larger allocation so nothing can fault.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint32_t *buf, const int *offs, int n, uint32_t *out4, uint32_t *out1) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(buf + 1024), 0, 64, 0x00020000);
  const int t = threadIdx.x;
  if (t < n) {
    u32x4 q;
    const int vo = offs[t];
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(vo), "s"(r) : "memory");
    for (int e = 0; e < 4; e++) out4[4 * t + e] = q[e];
    uint32_t d;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(vo), "s"(r) : "memory");
    out1[t] = d;
  }
}
int main() {
  const int N = 1 << 18;
  uint32_t *h = new uint32_t[N];
  for (int i = 0; i < N; i++) h[i] = 100000 + i;  // element i of the allocation; descriptor element j = 101024 + j
  uint32_t *d, *o4, *o1; int *doffs;
  int offs[] = {-8, -4, 0, 4, 8, 44, 48, 52, 56, 60, 64, 68, 1 << 16};
  const int n = sizeof(offs) / sizeof(int);
  hipMalloc(&d, N * 4); hipMalloc(&o4, n * 16); hipMalloc(&o1, n * 4); hipMalloc(&doffs, n * 4);
  hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice); hipMemcpy(doffs, offs, n * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, doffs, n, o4, o1);
  uint32_t r4[64], r1[16];
  hipMemcpy(r4, o4, n * 16, hipMemcpyDeviceToHost); hipMemcpy(r1, o1, n * 4, hipMemcpyDeviceToHost);
  printf("descriptor: base = element 1024 (value 101024), num_records = 64 bytes (16 dwords: 101024..101039)\n");
  for (int i = 0; i < n; i++)
    printf("voffset %6d: x4 = %u %u %u %u   dword = %u\n", offs[i], r4[4 * i], r4[4 * i + 1], r4[4 * i + 2], r4[4 * i + 3], r1[i]);
  return 0;
}
