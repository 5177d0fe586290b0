// Is the SCALAR offset of a raw buffer access part of gfx950's bounds check?  Descriptor in the middle of a larger allocation
// (nothing can fault): num_records = 64 bytes, loads with voffset x soffset combinations.  "excluded" = the check is
// voffset + 4 <= num_records whatever soffset is; "included" = voffset + soffset + 4 <= num_records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t *buf, uint32_t *out, uint32_t *st) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(buf + 1024), 0, 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(st + 1024, 0, 64, 0x00020000);
  const int t = threadIdx.x;          // voffset = 4 * t: lanes 0..15 inside num_records, 16.. outside
  const int vo = 4 * t;
  uint32_t d0, d1, d2, d3;
  int s0 = 0, s1 = 32, s2 = 64, s3 = 4096;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d0) : "v"(vo), "s"(r), "s"(s0) : "memory");
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d1) : "v"(vo), "s"(r), "s"(s1) : "memory");
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d2) : "v"(vo), "s"(r), "s"(s2) : "memory");
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d3) : "v"(vo), "s"(r), "s"(s3) : "memory");
  out[4 * t + 0] = d0; out[4 * t + 1] = d1; out[4 * t + 2] = d2; out[4 * t + 3] = d3;
  uint32_t val = 7000 + t;
  asm volatile("buffer_store_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" :: "v"(val), "v"(vo), "s"(w), "s"(s2) : "memory");
}
int main() {
  const int N = 1 << 16;
  uint32_t *h = new uint32_t[N];
  for (int i = 0; i < N; i++) h[i] = 100000 + i;
  uint32_t *d, *o, *st;
  hipMalloc(&d, N * 4); hipMalloc(&o, 64 * 16); hipMalloc(&st, N * 4);
  hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice); hipMemset(st, 0, N * 4);
  probe<<<1, 64>>>(d, o, st);
  uint32_t r[256], sb[N];
  hipMemcpy(r, o, 64 * 16, hipMemcpyDeviceToHost); hipMemcpy(sb, st, N * 4, hipMemcpyDeviceToHost);
  printf("descriptor: base = element 1024 (value 101024), num_records = 64 bytes; voffset = 4 * lane\n");
  for (int t = 0; t < 24; t++)
    printf("lane %2d voffset %3d: soffset 0 -> %u | 32 -> %u | 64 -> %u | 4096 -> %u\n", t, 4 * t, r[4 * t], r[4 * t + 1], r[4 * t + 2], r[4 * t + 3]);
  int written = 0, first = -1, last = -1;
  for (int i = 0; i < N; i++) if (sb[i]) { written++; if (first < 0) first = i; last = i; }
  printf("stores with soffset 64: %d dwords written, elements %d .. %d of the allocation (descriptor base = 1024; +16 = soffset applied)\n", written, first, last);
  const bool excl = r[4 * 8 + 1] == 101024 + 8 + 8 && r[4 * 15 + 2] == 101024 + 15 + 16;
  const bool incl = r[4 * 8 + 1] == 0;
  printf("VERDICT: soffset is %s the bounds check\n", excl ? "EXCLUDED from" : incl ? "INCLUDED in" : "?? (neither pattern)");
  return 0;
}
