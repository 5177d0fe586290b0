// laser_amd/csrc/capi.cpp -- the extern "C" boundary of liblaser_hip.so (include/laser_hip.h).
//
// Host-pointer entry points reproduce Laser's blocking call semantics (caller owns every pointer,
// the call returns when C is final); `_dev` entry points are the device-resident, stream-ordered
// path that benchmarks and multi-GPU code use.  No CPU compute fallback exists anywhere in this
// library: without a usable gfx950 device every compute entry point returns LASER_HIP_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/laser_hip.h"
#include "capi_internal.h"
#include "common.h"

using namespace laser_hip;

namespace {

thread_local std::string g_err;
std::mutex g_mu;  // guards initialisation and the registries (pre-pack handles, storage free list, host-range registry)

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      const char *why_ = asm_error_detail();                                                  \
      return fail(LASER_HIP_E_HIP, "%s failed: %s%s%s (%s:%d)", #expr, hipGetErrorString(e_), \
                  why_[0] ? ": " : "", why_, __FILE__, __LINE__);                             \
    }                                                                                         \
  } while (0)

struct Context {
  std::atomic<bool> ready{false};  // read lock-free on every call, written once under g_mu
  int device = -1;
  std::string arch;
  // run-time knobs: set by one thread, read by every call on every thread -> relaxed atomics (a knob flipped while a
  // call is in flight applies to that call or the next, never tears)
  std::atomic<int> float_mode{LASER_HIP_F32_LASER_ORDER};
  std::atomic<int> f32_cfg{-1};
  hipStream_t s_up = nullptr, s_comp = nullptr;  // host-pointer pipeline: uploads / kernels
  std::atomic<bool> f64_mfma{true};       // float64 GEMM on the f64 matrix cores (false: VALU kernel)
  std::atomic<bool> i32_mfma{true};       // int32 GEMM on the int8 matrix cores (false: VALU kernel)
  std::atomic<bool> i64_mfma{true};       // int64 GEMM on the int8 matrix cores (false: VALU kernel)
  std::atomic<bool> zc_poll{true};           // small host-pointer calls: poll completion flags in mapped memory (false: synchronise the stream)
  std::atomic<bool> host_pipeline_2d{true};  // large row-major host-pointer calls: row panels x column panels (false: row panels only)
  std::atomic<int> slice_parallel_min{2};        // (tuning override only) fewest kc slices worth splitting
  std::atomic<int64_t> slice_parallel_tiles{0};  // tuning override of the tile-count limit of the slice-parallel form (0 = built-in rule)
  std::atomic<bool> slice_parallel{true}; // few tiles x long K: kc slices as one batched launch + ordered combine
  std::atomic<bool> skinny{true};         // M <= 8 or N <= 8: the streaming kernel (false: always the tiled kernels)
  std::atomic<bool> conv_implicit{true};  // fuse im2col into the GEMM's B loader (false: explicit workspace)
  std::atomic<int> shard_devices{1};      // host-pointer gemm_strided: row panels over this many GPUs (1 = off; laser_hip_set_shard_devices)
};
Context g_ctx;

// Per-DEVICE state of the host-pointer paths: cached scratch (one growing buffer per role), the streams of the
// upload / compute / download pipeline, and a mutex that serialises the host-pointer calls using THIS device.  One
// process may drive every GPU of the node (the sharded entry points run one host thread per device), so none of this
// is per process.
constexpr int kMaxDevices = 16;
struct DeviceCtx {
  int device = -1;
  std::mutex mu;
  hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
  void *scratch[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t scratch_sz[6] = {0, 0, 0, 0, 0, 0};
  uint32_t zc_seq = 0;  // sequence number of the zero-copy path's completion flags (never 0 in a flag)
  void *zc = nullptr;  // pinned host buffer mapped into the device: zero-copy staging of the small-problem host path
  size_t zc_sz = 0;
};
DeviceCtx g_dev[kMaxDevices];
// The device a host-pointer call on this thread uses: -1 = the library's default device (laser_hip_init); the
// sharded entry points set it in their per-device worker threads.
thread_local int tl_device = -1;
thread_local int tl_f32_cfg = -2;  // per-thread override of g_ctx.f32_cfg (-2 = none), api_set_thread_f32_config
inline int f32_cfg_now() { return tl_f32_cfg >= -1 ? tl_f32_cfg : g_ctx.f32_cfg.load(); }
thread_local DeviceCtx *tl_dev = nullptr;  // valid while a HostCall guard is alive

int ensure_init_locked(int device) {
  if (g_ctx.ready && (device < 0 || device == g_ctx.device)) return LASER_HIP_OK;
  // (a later init with another ordinal only moves the DEFAULT device of the host-pointer entry points: scratch,
  // streams and kernel attributes are kept per device, nothing of the previous device is reused)
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(LASER_HIP_E_NODEVICE, "no HIP device available (%s)",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0) {
    HIP_TRY(hipGetDevice(&device));
  } else {
    if (device >= n || device >= kMaxDevices) return fail(LASER_HIP_E_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  std::string arch = prop.gcnArchName;
  // gcnArchName looks like "gfx950:sramecc+:xnack-"; this library carries gfx950 code objects only
  if (arch.rfind("gfx950", 0) != 0)
    return fail(LASER_HIP_E_NODEVICE, "device %d is %s; liblaser_hip is built for gfx950 (MI355X) only",
                device, arch.c_str());
  g_ctx.arch = arch.substr(0, arch.find(':'));
  g_ctx.device = device;
  g_ctx.ready = true;
  return LASER_HIP_OK;
}

int ensure_init() {
  if (g_ctx.ready) return LASER_HIP_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  return ensure_init_locked(-1);
}

// RAII guard of a host-pointer entry point: makes its device current for the calling thread (another host thread
// using the host-pointer API would otherwise run on device 0 with this device's buffers), serialises on that device's
// mutex and publishes its DeviceCtx in tl_dev.
struct HostCall {
  int rc = LASER_HIP_OK;
  DeviceCtx *d = nullptr;
  DeviceCtx *prev = nullptr;
  int caller_dev = -1;  // the caller's current device, restored on exit: an application thread driving another GPU must
                        // not find itself on the library's device after a host-pointer call
  HostCall() {
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
    const int dev = tl_device >= 0 ? tl_device : g_ctx.device;
    if (dev < 0 || dev >= kMaxDevices) {
      rc = fail(LASER_HIP_E_INVALID, "device %d outside 0..%d", dev, kMaxDevices - 1);
      return;
    }
    const hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) {
      rc = fail(LASER_HIP_E_HIP, "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
      return;
    }
    d = &g_dev[dev];
    d->mu.lock();
    d->device = dev;
    prev = tl_dev;
    tl_dev = d;
  }
  ~HostCall() {
    if (d) {
      tl_dev = prev;
      d->mu.unlock();
    }
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
  }
  HostCall(const HostCall &) = delete;
  HostCall &operator=(const HostCall &) = delete;
};

int scratch_get(int slot, size_t bytes, void **out) {
  DeviceCtx &D = *tl_dev;
  if (bytes == 0) bytes = 16;
  if (D.scratch_sz[slot] < bytes) {
    if (D.scratch[slot]) HIP_TRY(hipFree(D.scratch[slot]));
    D.scratch[slot] = nullptr;
    D.scratch_sz[slot] = 0;
    size_t want = bytes + bytes / 8;  // a little headroom so slowly growing sizes do not thrash
    HIP_TRY(hipMalloc(&D.scratch[slot], want));
    D.scratch_sz[slot] = want;
  }
  *out = D.scratch[slot];
  return LASER_HIP_OK;
}

// Pinned, device-mapped host staging for small host-pointer problems (<= kZeroCopyMax bytes of operands): the kernel
// reads A and B from it and writes C into it across PCIe -- one round trip, because the small-matrix kernel issues all
// of a block's loads up front -- instead of three blocking hipMemcpy calls (~10-15 us each for 64 KiB).
constexpr size_t kZeroCopyMax = (size_t)1 << 20;
int zero_copy_get(size_t bytes, void **out) {
  DeviceCtx &D = *tl_dev;
  if (D.zc_sz < bytes) {
    if (D.zc) HIP_TRY(hipHostFree(D.zc));
    D.zc = nullptr;
    D.zc_sz = 0;
    const size_t want = std::max<size_t>(bytes, (size_t)256 << 10);
    HIP_TRY(hipHostMalloc(&D.zc, want, hipHostMallocDefault));
    memset(D.zc, 0, want);  // completion flags live here: a fresh buffer must not hold a value that looks like a sequence number
    D.zc_sz = want;
  }
  *out = D.zc;
  return LASER_HIP_OK;
}

int pipeline_streams() {
  DeviceCtx &D = *tl_dev;
  if (!D.s_up) {
    HIP_TRY(hipStreamCreateWithFlags(&D.s_up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&D.s_comp, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&D.s_down, hipStreamNonBlocking));
  }
  return LASER_HIP_OK;
}

// Lowest / highest element offset touched by an R x C strided view (strides may be negative).
void view_span(int64_t R, int64_t C, int64_t rs, int64_t cs, int64_t *lo, int64_t *hi) {
  const int64_t r = (R - 1) * rs, c = (C - 1) * cs;
  *lo = std::min<int64_t>(0, r) + std::min<int64_t>(0, c);
  *hi = std::max<int64_t>(0, r) + std::max<int64_t>(0, c);
}

hipError_t launch_tiled(const GemmArgs<float> &a, hipStream_t s) {
  if (f32_cfg_now() < 0) {   // (batched: the hand-scheduled kernels take the batch index as grid y)
    const hipError_t e = launch_gemm_f32_asm(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
    if (e != hipErrorNotSupported) return e;
  }
  return launch_gemm_f32(a, f32_cfg_now(), g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
}
hipError_t launch_tiled(const GemmArgs<double> &a, hipStream_t s) {
  if (g_ctx.f64_mfma) {
    const hipError_t e = launch_gemm_f64_asm(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
    if (e != hipErrorNotSupported) return e;
  }
  return launch_gemm_f64(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
}

// Slice-parallel GEMM for problems with few output tiles and a long K (tall-skinny products, small M x N with a huge
// K): Laser's kc slices are independent chains from +0 -- only their sums are added in order (gemm.nim:150-158) -- so
// the slices are computed as ONE batched launch (batch = slice, K = kc, each a single-chain product into a workspace
// W[p][M][N]) and folded by an ordered combine pass.  Same fused multiply-adds in the same order, same unfused
// alpha / beta arithmetic => bit-identical to the sequential kernel, with ceil(K / kc) times the workgroups.
// Returns hipErrorNotSupported when the shape does not call for it.
template <typename T>
hipError_t gemm_slice_parallel(const GemmArgs<T> &a, int kc, hipStream_t s) {
  if (!g_ctx.slice_parallel || a.batch != 1 || a.bias != nullptr || a.act != 0) return hipErrorNotSupported;
  if (a.M <= 0 || a.N <= 0 || a.K <= kc) return hipErrorNotSupported;
  const int64_t tiles64 = ((a.M + 63) / 64) * ((a.N + 63) / 64);
  const int64_t nfull = a.K / kc, nsl = (a.K + kc - 1) / kc;
  const double ws_bytes = (double)nsl * (double)a.M * (double)a.N * sizeof(T);
  // measured boundary (scripts/slice_parallel_threshold_probe.py, slice_parallel_min_probe.py): the fewer the tiles the
  // fewer slices it takes to pay -- up to ~150 tiles of 64x64 from two slices on (768^2 x 1536: +20 %), up to ~400 from
  // five (1280^2 x 2560: +17 %; 1024^2 x 2048 with four: -3 %), up to ~600 from six (1536^2 x 6144: +24 %); from ~1000
  // tiles on the sequential loop wins (2048^3: -14 %)
  int64_t need = tiles64 <= 150 ? 2 : tiles64 <= 400 ? 5 : tiles64 <= 600 ? 6 : (int64_t)1 << 40;
  if (a.K % kc != 0 && need < 3) need = 3;  // a ragged last slice is a launch of its own: 768^3 (512 + 256) loses
  if (g_ctx.slice_parallel_tiles > 0) need = tiles64 <= g_ctx.slice_parallel_tiles.load() ? (int64_t)g_ctx.slice_parallel_min.load() : (int64_t)1 << 40;  // tuning override
  if (nsl < need || nsl > 65535 || ws_bytes > 1.5e9) return hipErrorNotSupported;
  T *W = nullptr;
  hipError_t e = scratch_alloc_async((void **)&W, (size_t)ws_bytes, s);
  if (e != hipSuccess) return e;
  const int64_t mn = a.M * a.N;
  GemmArgs<T> b = a;
  b.alpha = (T)1; b.beta = (T)0;
  b.C = W; b.rsC = a.N; b.csC = 1; b.bsC = mn;
  b.K = kc; b.Kext = kc;
  b.batch = (int32_t)nfull;
  b.bsA = (int64_t)kc * a.csA;  // slice p starts kc columns of A / rows of B further on
  b.bsB = (int64_t)kc * a.rsB;
  e = launch_tiled(b, s);
  if (e == hipSuccess && nsl > nfull) {  // the ragged last slice
    GemmArgs<T> c = b;
    c.batch = 1;
    c.K = a.K - nfull * kc; c.Kext = c.K;
    c.A = a.A + nfull * b.bsA;
    c.B = a.B + nfull * b.bsB;
    c.C = W + nfull * mn;
    e = launch_tiled(c, s);
  }
  if (e == hipSuccess) e = launch_combine_slices<T>(a.C, a.rsC, a.csC, W, a.M, a.N, (int)nsl, a.alpha, a.beta, s);
  const hipError_t e2 = hipFreeAsync(W, s);
  return e != hipSuccess ? e : e2;
}

template <typename T>
hipError_t run_gemm(const GemmArgs<T> &a, hipStream_t s);
template <typename T>
hipError_t run_gemm_core(const GemmArgs<T> &a, hipStream_t s);
template <>
hipError_t run_gemm_core<float>(const GemmArgs<float> &a, hipStream_t s);
template <>
hipError_t run_gemm_core<double>(const GemmArgs<double> &a, hipStream_t s);

// Ragged-by-a-few problems (4100^3, 4095 x 4097 x 4099): 1..8 rows / columns past a multiple of 64 cost a whole extra row /
// column of tiles (4100 = 16 x 256 + 4: 17 tile rows for 16.02 tile rows of work).  Elements of C are independent and the
// streaming kernel for M <= 8 or N <= 8 (gemm_skinny.hip) runs the same k-ascending, kc-sliced chain per element as the tiled
// kernels, so those few rows / columns are peeled off and streamed (HBM-bound, ~20 us each at 4100^3), and the tiled
// launch sees whole tiles: bit-identical to the single launch.  Laser-order arithmetic only (the streaming kernel always
// restarts its chain every kc), i.e. laser-order mode or K <= kc; plain (unfused, unbatched, not pre-packed) problems.
template <typename T>
hipError_t run_gemm_peeled(const GemmArgs<T> &a, int kc, bool laser, bool *taken, hipStream_t s) {
  *taken = false;
  if (!g_ctx.skinny || !g_split_tail || a.batch != 1 || a.bias != nullptr || a.act != 0) return hipSuccess;
  if (!(laser || a.K <= kc) || a.Mext != a.M || a.Next != a.N || a.K < 256) return hipSuccess;
  const int64_t rM = a.M % 64, rN = a.N % 64;
  const bool peel_m = rM >= 1 && rM <= 8 && a.M >= 1024 && a.N >= 512;
  const bool peel_n = rN >= 1 && rN <= 8 && a.N >= 1024 && a.M - (peel_m ? rM : 0) >= 512;
  if (!peel_m && !peel_n) return hipSuccess;
  *taken = true;
  const int64_t M1 = peel_m ? a.M - rM : a.M, N1 = peel_n ? a.N - rN : a.N;
  GemmArgs<T> m = a;  // whole tiles
  m.M = M1; m.Mext = M1; m.N = N1; m.Next = N1;
  hipError_t e = run_gemm_core<T>(m, s);
  if (e == hipSuccess && peel_n) {  // columns [N1, N) of rows [0, M1)
    GemmArgs<T> r = a;
    r.M = M1; r.Mext = M1; r.N = rN; r.Next = rN;
    r.B = a.B + N1 * a.csB;
    r.C = a.C + N1 * a.csC;
    e = launch_gemm_skinny<T>(r, true, kc, s);
  }
  if (e == hipSuccess && peel_m) {  // rows [M1, M), every column
    GemmArgs<T> b = a;
    b.M = rM; b.Mext = rM;
    b.A = a.A + M1 * a.rsA;
    b.C = a.C + M1 * a.rsC;
    e = launch_gemm_skinny<T>(b, true, kc, s);
  }
  return e == hipErrorNotSupported ? hipErrorInvalidValue : e;
}

// Fused prologue on a path that has no kernel for it: the operand is materialised once (relu applied while it is copied into a
// dense row-major scratch matrix -- "during the prepacking", README.md:243-244) and the plain problem runs on it.
template <typename T>
hipError_t run_gemm_prologue_materialised(const GemmArgs<T> &a, hipStream_t s) {
  if (a.batch != 1 || a.Mext != a.M || a.Next != a.N || a.Kext != a.K) return hipErrorInvalidValue;
  const size_t nA = a.preA ? (size_t)a.M * a.K : 0, nB = a.preB ? (size_t)a.K * a.N : 0;
  if (nA + nB == 0) return hipErrorInvalidValue;
  T *scratch = nullptr;
  hipError_t e = scratch_alloc_async((void **)&scratch, (nA + nB) * sizeof(T), s);
  if (e != hipSuccess) return e;
  GemmArgs<T> b = a;
  b.preA = b.preB = 0;
  if (a.preA) {
    e = launch_pack_pad<T>(scratch, a.M, a.K, a.A, a.M, a.K, a.rsA, a.csA, s, 1);
    b.A = scratch; b.rsA = a.K; b.csA = 1;
  }
  if (e == hipSuccess && a.preB) {
    e = launch_pack_pad<T>(scratch + nA, a.K, a.N, a.B, a.K, a.N, a.rsB, a.csB, s, 1);
    b.B = scratch + nA; b.rsB = a.N; b.csB = 1;
  }
  if (e == hipSuccess) e = run_gemm<T>(b, s);
  const hipError_t e2 = hipFreeAsync(scratch, s);
  return e != hipSuccess ? e : e2;
}

template <>
hipError_t run_gemm<float>(const GemmArgs<float> &a, hipStream_t s) {
  if (a.preA || a.preB) {      // the `_pre` assembly kernels (relu in the staging registers), else one materialising pass
    if (f32_cfg_now() < 0) {
      const hipError_t e = launch_gemm_f32_asm(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
      if (e != hipErrorNotSupported) return e;
    }
    g_last_f32_asm = 0;
    return run_gemm_prologue_materialised<float>(a, s);
  }
  if (f32_cfg_now() < 0) {
    bool taken;
    const hipError_t e = run_gemm_peeled<float>(a, 512, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, &taken, s);
    if (taken) return e;
  }
  return run_gemm_core<float>(a, s);
}
template <>
hipError_t run_gemm<double>(const GemmArgs<double> &a, hipStream_t s) {
  if (a.preA || a.preB) return run_gemm_prologue_materialised<double>(a, s);
  if (g_ctx.f64_mfma) {
    bool taken;
    const hipError_t e = run_gemm_peeled<double>(a, 256, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, &taken, s);
    if (taken) return e;
  }
  return run_gemm_core<double>(a, s);
}
template <>
hipError_t run_gemm_core<float>(const GemmArgs<float> &a, hipStream_t s) {
  if (f32_cfg_now() < 0 && g_ctx.skinny) {  // matrix-vector-like shapes: an HBM stream, not a tile problem
    const hipError_t e = launch_gemm_skinny<float>(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, 512, s);
    if (e != hipErrorNotSupported) return e;
  }
  if (f32_cfg_now() < 0) {  // few 32x32 blocks / batches of tiny matrices: one wave per block, no LDS round trips
    const hipError_t e = launch_gemm_small<float>(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, 512, s);
    if (e != hipErrorNotSupported) return e;
  }
  // Up to ~150 tiles of 64x64 the slice-parallel form (kc slices as one batched launch + ordered combine) fills the chip
  // better than any single launch; above that the hand-scheduled assembly kernels come first (their 64x64 tile covers the
  // few-tile x long-K problems the slice-parallel form was built for: 1024^2 x 8192 = 256 tiles).
  // (a pinned assembly tile class -- option "asm_tile", the sharded entry points' LASER_HIP_SHARD_PIN_TILE -- asks for THAT kernel
  // family: the slice-parallel form does not come first then)
  const bool few_tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64) <= 150 && asm_tile_pin_now() < 0;
  g_last_f32_asm = 0;
  if (f32_cfg_now() < 0 && few_tiles) {
    const hipError_t e = gemm_slice_parallel<float>(a, 512, s);
    if (e != hipErrorNotSupported) return e;
  }
  // (a badly filled last round of tiles is the assembly launcher's business: its persistent plan hands the chip's workgroup slots
  // equal numbers of kc slices and finishes a cut tile with an in-kernel ordered fix-up -- gemm_f32_asm.cpp plan_launch)
  if (f32_cfg_now() < 0) {
    const hipError_t e = launch_gemm_f32_asm(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
    if (e != hipErrorNotSupported) return e;
  }
  if (f32_cfg_now() < 0 && !few_tiles) {
    const hipError_t e = gemm_slice_parallel<float>(a, 512, s);
    if (e != hipErrorNotSupported) return e;
  }
  return launch_gemm_f32(a, f32_cfg_now(), g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s);
}
template <>
hipError_t run_gemm_core<double>(const GemmArgs<double> &a, hipStream_t s) {
  const bool laser = g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER;
  if (g_ctx.skinny) {
    const hipError_t e = launch_gemm_skinny<double>(a, laser, 256, s);
    if (e != hipErrorNotSupported) return e;
  }
  if (g_ctx.f64_mfma) {
    const hipError_t es = launch_gemm_small<double>(a, laser, 256, s);
    if (es != hipErrorNotSupported) return es;
    const bool few_tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64) <= 150;   // (same rule as float32)
    g_last_f64_asm = 0;
    g_last_split = 0;
    if (few_tiles) {
      const hipError_t e = gemm_slice_parallel<double>(a, 256, s);
      if (e != hipErrorNotSupported) return e;
    }
    const hipError_t ea = launch_gemm_f64_asm(a, laser, s);   // the hand-scheduled assembly kernels (laser_amd/asmgen/f64_kernel.py)
    if (ea != hipErrorNotSupported) return ea;
    if (!few_tiles) {
      const hipError_t e = gemm_slice_parallel<double>(a, 256, s);
      if (e != hipErrorNotSupported) return e;
    }
    return launch_gemm_f64(a, laser, s);
  }  // v_mfma_f64_16x16x4_f64: a k-ordered fma chain
  return launch_gemm_valu<double>(a, laser, s);
}
// Integer K beyond the hand-scheduled limb kernels' 8192 (their accumulator groups are never folded): arithmetic mod 2^n is
// associative, so C = alpha * sum_chunks(A_c B_c) + beta * C0 is computed chunk by chunk -- the first with (alpha, beta), the
// rest with (alpha, 1) -- bit for bit the single product.  hipErrorNotSupported (nothing launched): not the kernels' class.
template <typename T, typename FA, typename FC>
hipError_t int_gemm_k_chunks(const GemmArgs<T> &a, void *ws, hipStream_t s, FA asm_launch, FC compiler_launch) {
  constexpr int64_t kChunk = 8192;
  hipError_t e = hipErrorNotSupported;
  for (int64_t k0 = 0; k0 < a.K; k0 += kChunk) {
    GemmArgs<T> c = a;
    c.K = std::min(kChunk, a.K - k0);
    c.Kext = c.K;
    c.A = a.A + k0 * a.csA;
    c.B = a.B + k0 * a.rsB;
    if (k0 > 0) c.beta = (T)1;
    e = asm_launch(c, ws, s);
    if (e == hipErrorNotSupported) {
      if (k0 == 0) return e;
      e = compiler_launch(c, ws, s);
    }
    if (e != hipSuccess) return e;
  }
  return e;
}

template <>
hipError_t run_gemm<int32_t>(const GemmArgs<int32_t> &a, hipStream_t s) {
  // Large single problems go to the int8 matrix cores (limb decomposition, gemm_i32_mfma.hip); the
  // limb planes live in stream-ordered scratch so concurrent streams never share a buffer.
  if (g_ctx.skinny) {
    const hipError_t e = launch_gemm_skinny<int32_t>(a, false, 512, s);
    if (e != hipErrorNotSupported) return e;
  }
  const double work = (double)a.M * (double)a.N * (double)a.K;
  if (g_ctx.i32_mfma && a.batch == 1 && work >= 64.0 * 64.0 * 64.0 * 8.0) {
    void *ws = nullptr;
    hipError_t e = scratch_alloc_async(&ws, gemm_i32_mfma_workspace_bytes(a.M, a.N, a.K), s);
    if (e != hipSuccess) return e;
    g_last_i32_asm = 0;
    // the hand-scheduled kernel (laser_amd/asmgen/i8_kernel.py) when eligible, K > 8192 in chunks
    e = a.K > 8192 ? int_gemm_k_chunks<int32_t>(a, ws, s, launch_gemm_i32_asm, launch_gemm_i32_mfma) : launch_gemm_i32_asm(a, ws, s);
    if (e == hipErrorNotSupported) e = launch_gemm_i32_mfma(a, ws, s);
    hipError_t e2 = hipFreeAsync(ws, s);
    return e != hipSuccess ? e : e2;
  }
  return launch_gemm_valu<int32_t>(a, false, s);
}
template <>
hipError_t run_gemm<int64_t>(const GemmArgs<int64_t> &a, hipStream_t s) {
  if (g_ctx.skinny) {
    const hipError_t e = launch_gemm_skinny<int64_t>(a, false, 256, s);
    if (e != hipErrorNotSupported) return e;
  }
  // Large single problems: eight int8 limbs on the matrix cores (gemm_i64_mfma.hip), planes in stream-ordered scratch
  const double work = (double)a.M * (double)a.N * (double)a.K;
  if (g_ctx.i64_mfma && a.batch == 1 && work >= 64.0 * 64.0 * 64.0 * 8.0) {
    void *ws = nullptr;
    hipError_t e = scratch_alloc_async(&ws, gemm_i64_mfma_workspace_bytes(a.M, a.N, a.K), s);
    if (e != hipSuccess) return e;
    g_last_i32_asm = 0;
    // the hand-scheduled kernel (i8_kernel.py "i64_64x64x32") when eligible, K > 8192 in chunks
    e = a.K > 8192 ? int_gemm_k_chunks<int64_t>(a, ws, s, launch_gemm_i64_asm, launch_gemm_i64_mfma) : launch_gemm_i64_asm(a, ws, s);
    if (e == hipErrorNotSupported) e = launch_gemm_i64_mfma(a, ws, s);
    hipError_t e2 = hipFreeAsync(ws, s);
    return e != hipSuccess ? e : e2;
  }
  return launch_gemm_valu<int64_t>(a, false, s);
}

// the small-matrix kernel on operands that live in host memory mapped into the device (gemm_host's zero-copy staging)
template <typename T>
hipError_t run_small_mapped(const GemmArgs<T> &a, hipStream_t s) {
  if constexpr (std::is_same<T, float>::value)
    return launch_gemm_small<float>(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, 512, s, true);
  else if constexpr (std::is_same<T, double>::value)
    return launch_gemm_small<double>(a, g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, 256, s, true);
  else
    return hipErrorNotSupported;
}

template <typename T>
GemmArgs<T> make_args(int64_t batch, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA,
                      int64_t csA, int64_t bsA, const T *B, int64_t rsB, int64_t csB, int64_t bsB, T beta,
                      T *C, int64_t rsC, int64_t csC, int64_t bsC) {
  GemmArgs<T> a;
  memset(&a, 0, sizeof a);
  a.M = M; a.N = N; a.K = K;
  a.alpha = alpha; a.beta = beta;
  a.A = A; a.rsA = rsA; a.csA = csA; a.bsA = bsA;
  a.B = B; a.rsB = rsB; a.csB = csB; a.bsB = bsB;
  a.C = C; a.rsC = rsC; a.csC = csC; a.bsC = bsC;
  a.Mext = M; a.Next = N; a.Kext = K;
  a.batch = (int32_t)batch;
  return a;
}

// Fused epilogue request (device pointers): bias == nullptr and act == 0 means "plain gemm_strided".
template <typename T>
struct Epi {
  const T *bias = nullptr;
  int64_t rs = 0, cs = 0, bs = 0;
  int act = 0;
  int pre = 0;     // fused prologue: bit 0 = relu on A's elements, bit 1 = relu on B's (LASER_HIP_PRE_RELU_A / _B >> 8)
  bool on() const { return bias != nullptr || act != 0 || pre != 0; }
};
// the `activation` argument of the _ex_ entry points: low byte = activation, bits 8 / 9 = fused prologue
template <typename T>
Epi<T> make_epi(const T *bias, int64_t rs, int64_t cs, int act) {
  Epi<T> e;
  e.bias = bias; e.rs = rs; e.cs = cs;
  e.act = act & 0xff;
  e.pre = (act >> 8) & 3;
  if (act & ~0x3ff) e.act = -1;      // unknown bits: rejected by epi_check
  return e;
}
template <typename T>
int epi_check(const Epi<T> *e) {
  if (!e) return LASER_HIP_OK;
  if (e->act < LASER_HIP_ACT_NONE || e->act > LASER_HIP_ACT_SIGMOID) return fail(LASER_HIP_E_INVALID, "unknown activation %d", e->act);
  if (!std::is_floating_point<T>::value) return fail(LASER_HIP_E_INVALID, "fused epilogue is float32/float64 only");
  if (std::is_same<T, double>::value && !g_ctx.f64_mfma && (e->bias || e->act))
    return fail(LASER_HIP_E_INVALID, "fused epilogue needs the f64 MFMA kernel (laser_hip_set_option(\"f64_mfma\", 1))");
  return LASER_HIP_OK;
}
template <typename T>
void epi_apply(GemmArgs<T> &a, const Epi<T> *e) {
  if (!e) return;
  a.bias = e->bias; a.rsBias = e->rs; a.csBias = e->cs; a.bsBias = e->bs; a.act = e->act;
  a.preA = e->pre & 1; a.preB = (e->pre >> 1) & 1;
}

template <typename T>
int gemm_dev(int64_t batch, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA,
             int64_t bsA, const T *B, int64_t rsB, int64_t csB, int64_t bsB, T beta, T *C, int64_t rsC,
             int64_t csC, int64_t bsC, void *stream, const Epi<T> *epi = nullptr) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = epi_check<T>(epi)) return rc;
  if (int rc = ensure_init()) return rc;
  // K == 0: the reference's pc loop never runs, C is left untouched even if beta != 1 (gemm.nim:150)
  if (M == 0 || N == 0 || K == 0 || batch == 0) return LASER_HIP_OK;
  if (!A || !B || !C) return fail(LASER_HIP_E_INVALID, "null operand pointer");
  if (batch > 65535) return fail(LASER_HIP_E_INVALID, "batch > 65535");
  GemmArgs<T> a = make_args<T>(batch, M, N, K, alpha, A, rsA, csA, bsA, B, rsB, csB, bsB, beta, C, rsC, csC, bsC);
  epi_apply(a, epi);
  HIP_TRY(run_gemm<T>(a, (hipStream_t)stream));
  return LASER_HIP_OK;
}

// Row-panel pipelined host path (see gemm_host).  dA0/dB0/dC0 are the device addresses of element
// (0,0) of each operand inside the cached scratch; Bspan/bn = host span of B; dBbuf = its device copy.
template <typename T>
int gemm_host_pipelined(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *Bspan,
                        size_t bn, int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC, const T *dA0,
                        const T *dB0, T *dBbuf, T *dC0, bool c_up) {
  if (int rc = pipeline_streams()) return rc;
  DeviceCtx &D = *tl_dev;
  int64_t R = (M / 8 + 255) / 256 * 256;  // ~8 panels, whole 256-row tiles
  if (R < 256) R = 256;
  const int nchunks = (int)((M + R - 1) / R);
  std::vector<hipEvent_t> ev_up(nchunks), ev_comp(nchunks);
  for (int i = 0; i < nchunks; i++) {
    HIP_TRY(hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_comp[i], hipEventDisableTiming));
  }
  const int64_t ca = (K - 1) * csA, cc = (N - 1) * csC;
  auto a_span = [&](int64_t r0, int64_t r1, int64_t *lo, int64_t *hi) {
    *lo = r0 * rsA + std::min<int64_t>(0, ca);
    *hi = (r1 - 1) * rsA + std::max<int64_t>(0, ca);
  };
  auto c_span = [&](int64_t r0, int64_t r1, int64_t *lo, int64_t *hi) {
    *lo = r0 * rsC + std::min<int64_t>(0, cc);
    *hi = (r1 - 1) * rsC + std::max<int64_t>(0, cc);
  };

  // helper thread: copy finished C panels back while the main thread keeps uploading
  std::mutex qm;
  std::condition_variable qcv;
  std::deque<int> queue;
  bool closed = false;
  hipError_t down_err = hipSuccess;
  const int device = D.device;
  const hipStream_t s_down = D.s_down;
  std::thread downloader([&]() {
    (void)hipSetDevice(device);
    for (;;) {
      int i;
      {
        std::unique_lock<std::mutex> lk(qm);
        qcv.wait(lk, [&] { return !queue.empty() || closed; });
        if (queue.empty()) return;
        i = queue.front();
        queue.pop_front();
      }
      const int64_t r0 = i * R, r1 = std::min<int64_t>(M, r0 + R);
      int64_t lo, hi;
      c_span(r0, r1, &lo, &hi);
      // (asynchronous copy on its own stream + synchronise, not a blocking hipMemcpy: beside the uploads of s_up the
      // blocking form gets 26 GB/s each way, this one 46 -- scripts/pcie_probe.py, pageable memory)
      hipError_t e = hipEventSynchronize(ev_comp[i]);
      if (e == hipSuccess) e = hipMemcpyAsync(C + lo, dC0 + lo, (size_t)(hi - lo + 1) * sizeof(T), hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess) e = hipStreamSynchronize(s_down);
      if (e != hipSuccess && down_err == hipSuccess) down_err = e;
    }
  });
  auto finish = [&]() {
    {
      std::lock_guard<std::mutex> lk(qm);
      closed = true;
    }
    qcv.notify_all();
    downloader.join();
    for (int i = 0; i < nchunks; i++) {
      (void)hipEventDestroy(ev_up[i]);
      (void)hipEventDestroy(ev_comp[i]);
    }
  };
#define PIPE_TRY(expr)                                                                                        \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      finish();                                                                                               \
      (void)hipDeviceSynchronize();                                                                           \
      return fail(LASER_HIP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                         \
  } while (0)

  PIPE_TRY(hipMemcpyAsync(dBbuf, Bspan, bn * sizeof(T), hipMemcpyHostToDevice, D.s_up));
  for (int i = 0; i < nchunks; i++) {
    const int64_t r0 = i * R, r1 = std::min<int64_t>(M, r0 + R);
    int64_t lo, hi;
    a_span(r0, r1, &lo, &hi);
    PIPE_TRY(hipMemcpyAsync((T *)dA0 + lo, A + lo, (size_t)(hi - lo + 1) * sizeof(T), hipMemcpyHostToDevice, D.s_up));
    if (c_up) {
      c_span(r0, r1, &lo, &hi);
      PIPE_TRY(hipMemcpyAsync(dC0 + lo, C + lo, (size_t)(hi - lo + 1) * sizeof(T), hipMemcpyHostToDevice, D.s_up));
    }
    PIPE_TRY(hipEventRecord(ev_up[i], D.s_up));
    PIPE_TRY(hipStreamWaitEvent(D.s_comp, ev_up[i], 0));
    GemmArgs<T> a = make_args<T>(1, r1 - r0, N, K, alpha, dA0 + r0 * rsA, rsA, csA, 0, dB0, rsB, csB, 0, beta,
                                 dC0 + r0 * rsC, rsC, csC, 0);
    PIPE_TRY(run_gemm<T>(a, D.s_comp));
    PIPE_TRY(hipEventRecord(ev_comp[i], D.s_comp));
    {
      std::lock_guard<std::mutex> lk(qm);
      queue.push_back(i);
    }
    qcv.notify_one();
  }
#undef PIPE_TRY
  finish();
  if (down_err != hipSuccess) return fail(LASER_HIP_E_HIP, "D2H of a C panel failed: %s", hipGetErrorString(down_err));
  return LASER_HIP_OK;
}

// 2-D pipelined host path: row panels of A x column panels of B.  The row-panel pipeline above cannot start a kernel
// before ALL of B has crossed PCIe (8192^3: 4.7 of its ~14 ms); here B is cut into column panels as well and the
// uploads grow the computable region as a square -- B_0, A_0, then whichever of the next A row panel / B column panel
// keeps (rows up)/M ~ (cols up)/N -- so the first tile multiplies after one panel of each, every upload releases a row
// or column STRIP of C as one launch (the new row panel x every column already up, or the new column panel x every row
// already up: 12 launches at 8192^3, growing with what is on the device -- one launch per TILE left three quarters of
// the chip idle, 17.7 ms vs 15.5 ms for the row-panel form), and every finished strip is copied back while later panels
// are still arriving.  Elements of C are independent and each is ONE chain over all of K, so the arithmetic is unchanged.  Needs row-major-like
// operands (unit column stride; the column panels of B and the tiles of C are pitched 2-D copies).
template <typename T>
int gemm_host_pipelined2d(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, const T *B, int64_t rsB, T beta,
                          T *C, int64_t rsC, const T *dA0, const T *dB0, T *dC0, bool c_up) {
  if (int rc = pipeline_streams()) return rc;
  DeviceCtx &D = *tl_dev;
  int64_t R = (M / 8 + 255) / 256 * 256;  // ~8 row panels of whole 256-row tiles
  if (R < 256) R = 256;
  const int nI = (int)((M + R - 1) / R);
  int64_t W = (N / 4 + 255) / 256 * 256;  // ~4 column panels of whole 256-column tiles
  if (W < 256) W = 256;
  const int nJ = (int)((N + W - 1) / W);
  // one strip of C per upload: the new row panel x every column already up, or the new column panel x every row already up
  struct Strip { int64_t r0, r1, c0, c1; hipEvent_t done; };
  std::vector<Strip> strips;
  strips.reserve((size_t)nI + nJ);
  std::vector<hipEvent_t> ev_up;
  std::mutex qm;
  std::condition_variable qcv;
  std::deque<int> queue;
  bool closed = false;
  hipError_t down_err = hipSuccess;
  const int device = D.device;
  const hipStream_t s_down = D.s_down;
  std::thread downloader([&]() {
    (void)hipSetDevice(device);
    for (;;) {
      Strip t;
      {
        std::unique_lock<std::mutex> lk(qm);
        qcv.wait(lk, [&] { return !queue.empty() || closed; });
        if (queue.empty()) return;
        t = strips[queue.front()];
        queue.pop_front();
      }
      hipError_t e = hipEventSynchronize(t.done);
      if (e == hipSuccess) {
        // (asynchronous copies on their own stream + synchronise: see gemm_host_pipelined)
        if (t.c0 == 0 && t.c1 == N)  // whole rows: one contiguous span
          e = hipMemcpyAsync(C + t.r0 * rsC, dC0 + t.r0 * rsC, (size_t)((t.r1 - t.r0 - 1) * rsC + N) * sizeof(T), hipMemcpyDeviceToHost, s_down);
        else
          e = hipMemcpy2DAsync(C + t.r0 * rsC + t.c0, (size_t)rsC * sizeof(T), dC0 + t.r0 * rsC + t.c0, (size_t)rsC * sizeof(T),
                               (size_t)(t.c1 - t.c0) * sizeof(T), (size_t)(t.r1 - t.r0), hipMemcpyDeviceToHost, s_down);
        if (e == hipSuccess) e = hipStreamSynchronize(s_down);
      }
      if (e != hipSuccess && down_err == hipSuccess) down_err = e;
    }
  });
  auto finish = [&]() {
    {
      std::lock_guard<std::mutex> lk(qm);
      closed = true;
    }
    qcv.notify_all();
    downloader.join();
    for (hipEvent_t e : ev_up) (void)hipEventDestroy(e);
    for (Strip &t : strips) (void)hipEventDestroy(t.done);
  };
#define PIPE_TRY(expr)                                                                                        \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      finish();                                                                                               \
      (void)hipDeviceSynchronize();                                                                           \
      return fail(LASER_HIP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                         \
  } while (0)
  // one upload (row panel of A [+ of C when it is read], or column panel of B), then the strip of C it completes
  int a_up = 0, b_up = 0;
  auto upload_and_launch = [&](bool is_a) -> int {
    hipEvent_t ev;
    PIPE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ev_up.push_back(ev);
    Strip t;
    if (is_a) {
      t.r0 = a_up * R; t.r1 = std::min<int64_t>(M, t.r0 + R);
      t.c0 = 0; t.c1 = std::min<int64_t>(N, b_up * W);
      PIPE_TRY(hipMemcpyAsync((T *)dA0 + t.r0 * rsA, A + t.r0 * rsA, (size_t)((t.r1 - t.r0 - 1) * rsA + K) * sizeof(T), hipMemcpyHostToDevice, D.s_up));
      if (c_up)
        PIPE_TRY(hipMemcpyAsync(dC0 + t.r0 * rsC, C + t.r0 * rsC, (size_t)((t.r1 - t.r0 - 1) * rsC + N) * sizeof(T), hipMemcpyHostToDevice, D.s_up));
      a_up++;
    } else {
      t.c0 = b_up * W; t.c1 = std::min<int64_t>(N, t.c0 + W);
      t.r0 = 0; t.r1 = std::min<int64_t>(M, a_up * R);
      PIPE_TRY(hipMemcpy2DAsync((T *)dB0 + t.c0, (size_t)rsB * sizeof(T), B + t.c0, (size_t)rsB * sizeof(T), (size_t)(t.c1 - t.c0) * sizeof(T),
                                (size_t)K, hipMemcpyHostToDevice, D.s_up));
      b_up++;
    }
    if (t.r1 <= t.r0 || t.c1 <= t.c0) return LASER_HIP_OK;  // the very first upload has no partner yet
    PIPE_TRY(hipEventRecord(ev, D.s_up));
    PIPE_TRY(hipStreamWaitEvent(D.s_comp, ev, 0));  // s_up is in order: this event covers every earlier upload too
    GemmArgs<T> a = make_args<T>(1, t.r1 - t.r0, t.c1 - t.c0, K, alpha, dA0 + t.r0 * rsA, rsA, 1, 0, dB0 + t.c0, rsB, 1, 0, beta,
                                 dC0 + t.r0 * rsC + t.c0, rsC, 1, 0);
    PIPE_TRY(run_gemm<T>(a, D.s_comp));
    PIPE_TRY(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    PIPE_TRY(hipEventRecord(t.done, D.s_comp));
    {
      std::lock_guard<std::mutex> lk(qm);
      strips.push_back(t);
      queue.push_back((int)strips.size() - 1);
    }
    qcv.notify_one();
    return LASER_HIP_OK;
  };
  while (a_up < nI || b_up < nJ) {
    // B first, then keep the uploaded fractions level (ties go to A: its panels are the cheaper, contiguous copies)
    const bool take_b = b_up < nJ && (a_up >= nI || b_up == 0 || (int64_t)b_up * nI < (int64_t)a_up * nJ);
    if (int rc = upload_and_launch(!take_b)) return rc;
  }
#undef PIPE_TRY
  finish();
  if (down_err != hipSuccess) return fail(LASER_HIP_E_HIP, "D2H of a C strip failed: %s", hipGetErrorString(down_err));
  return LASER_HIP_OK;
}

// Host-pointer gemm_strided: stage the touched span of each operand, run, copy the C span back.
template <typename T>
int gemm_host(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *B,
              int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC, const Epi<T> *hepi = nullptr) {
  if (M < 0 || N < 0 || K < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = epi_check<T>(hepi)) return rc;
  if (int rc = ensure_init()) return rc;
  if (M == 0 || N == 0 || K == 0) return LASER_HIP_OK;
  if (!A || !B || !C) return fail(LASER_HIP_E_INVALID, "null operand pointer");
  // laser_hip_set_shard_devices(n != 1): a large plain gemm_strided call is cut into row ranges over n GPUs (rows of
  // C are independent, gemm.nim:160-176, so the arithmetic is unchanged).  Not for calls that already ARE a shard
  // (tl_device set by the sharded entry point's worker thread) and not worth it below ~4 tile rows per GPU.
  if (g_ctx.shard_devices != 1 && tl_device < 0 && !(hepi && hepi->on())) {
    int ndev = g_ctx.shard_devices;
    if (ndev <= 0 && hipGetDeviceCount(&ndev) != hipSuccess) ndev = 1;
    if (ndev > 1 && M >= (int64_t)1024 * ndev && (double)M * (double)N * (double)K >= 64.0 * 1024 * 1024 * 1024)
      return api_sharded_host<T>(ndev, M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC);
  }
  HostCall hc;
  if (hc.rc) return hc.rc;
  int64_t alo, ahi, blo, bhi, clo, chi;
  view_span(M, K, rsA, csA, &alo, &ahi);
  view_span(K, N, rsB, csB, &blo, &bhi);
  view_span(M, N, rsC, csC, &clo, &chi);
  const size_t an = (size_t)(ahi - alo + 1), bn = (size_t)(bhi - blo + 1), cn = (size_t)(chi - clo + 1);
  // small problems (BASELINE configs[0], fp32 128^3): zero-copy through the pinned staging buffer
  {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t ab = up(an * sizeof(T)), bb = up(bn * sizeof(T)), cb = up(cn * sizeof(T));
    const bool has_epi = hepi && hepi->on();
    if (!has_epi && f32_cfg_now() < 0 && ab + bb + cb <= kZeroCopyMax && gemm_small_takes((int)sizeof(T), M, N, K, 1, true) &&
        std::is_floating_point<T>::value) {
      // completion flags, one per workgroup (= per 32x32 / 16x16 block of C; at most 256 by the dispatch rule), behind C
      const int mb = sizeof(T) == 4 ? 32 : 16;
      const size_t nblk = (size_t)((M + mb - 1) / mb) * (size_t)((N + mb - 1) / mb);
      const size_t fb = up(nblk * sizeof(uint32_t));
      void *z;
      if (int rc = zero_copy_get(ab + bb + cb + fb, &z)) return rc;
      if (int rc = pipeline_streams()) return rc;
      T *hA = (T *)z, *hB = (T *)((char *)z + ab), *hC = (T *)((char *)z + ab + bb);
      volatile uint32_t *flags = (volatile uint32_t *)((char *)z + ab + bb + cb);
      // the flag words sit where an earlier call of another shape kept payload: a stale word that happens to equal this
      // call's sequence number would read as "block done" -- clear them before the launch
      for (size_t i = 0; i < nblk; i++) flags[i] = 0;
      memcpy(hA, A + alo, an * sizeof(T));
      memcpy(hB, B + blo, bn * sizeof(T));
      const bool c_in = (beta != (T)0) || cn != (size_t)M * (size_t)N;  // read, or a span with gaps that belong to the caller
      if (c_in) memcpy(hC, C + clo, cn * sizeof(T));
      GemmArgs<T> a = make_args<T>(1, M, N, K, alpha, hA - alo, rsA, csA, 0, hB - blo, rsB, csB, 0, beta, hC - clo, rsC, csC, 0);
      uint32_t seq = ++tl_dev->zc_seq;
      if (seq == 0) seq = ++tl_dev->zc_seq;  // flags hold the sequence number of the call that set them; 0 = never
      if (g_ctx.zc_poll) {
        a.done_flags = (uint32_t *)flags;
        a.done_seq = seq;
      }
      HIP_TRY(run_small_mapped<T>(a, tl_dev->s_comp));
      // Wait for the blocks' flags in mapped memory instead of synchronising the stream (the runtime's completion path
      // costs more than this kernel).  Bounded: after ~2 ms of polling fall back to the synchronise, which also reports
      // a failed launch.
      bool polled = false;
      if (g_ctx.zc_poll) {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        size_t i = 0;
        for (unsigned spins = 0; i < nblk; spins++) {
          if (flags[i] == seq) { i++; continue; }
          if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() > t_end) break;
          __builtin_ia32_pause();
        }
        polled = i == nblk;
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      if (!polled) HIP_TRY(hipStreamSynchronize(tl_dev->s_comp));
      memcpy(C + clo, hC, cn * sizeof(T));
      return LASER_HIP_OK;
    }
  }
  void *dA, *dB, *dC;
  if (int rc = scratch_get(0, an * sizeof(T), &dA)) return rc;
  if (int rc = scratch_get(1, bn * sizeof(T), &dB)) return rc;
  if (int rc = scratch_get(2, cn * sizeof(T), &dC)) return rc;
  // C must be uploaded when it is read (beta != 0) or when its span has gaps that belong to the
  // caller (so the copy-back restores them unchanged).  A dense C with beta == 0 is write-only.
  const bool c_dense = (cn == (size_t)M * (size_t)N);
  const bool c_up = (beta != (T)0 || !c_dense);
  const T *dA0 = (const T *)dA - alo, *dB0 = (const T *)dB - blo;
  T *dC0 = (T *)dC - clo;

  // Large row-major-like problems: stream row panels so PCIe and the kernel overlap (rows of C are
  // independent, so the per-element arithmetic is unchanged):
  //   main thread : H2D B, then per panel  H2D A_i [+ C_i]  ->  kernel_i on its own stream
  //   helper thread: D2H C_i as soon as kernel_i is done (PCIe is full duplex)
  auto iabs = [](int64_t v) { return v < 0 ? -v : v; };
  const bool panels_disjoint = rsA > 0 && rsC > 0 && rsA >= iabs(csA) * (K - 1) + 1 && rsC >= iabs(csC) * (N - 1) + 1;
  // fused epilogue: hepi->bias is a HOST view here; its span goes to the device next to the operands
  Epi<T> depi;
  const bool fused = hepi && hepi->on();
  if (fused) {
    depi = *hepi;
    if (hepi->bias) {
      int64_t lo, hi;
      view_span(M, N, hepi->rs, hepi->cs, &lo, &hi);
      void *dbias;
      if (int rc = scratch_get(5, (size_t)(hi - lo + 1) * sizeof(T), &dbias)) return rc;
      HIP_TRY(hipMemcpy(dbias, hepi->bias + lo, (size_t)(hi - lo + 1) * sizeof(T), hipMemcpyHostToDevice));
      depi.bias = (const T *)dbias - lo;
    }
  }
  // both operands and C row-major-like, and big enough that B's upload is worth hiding: the 2-D form.  Pinned (or
  // registered) B and C only: their column panels / strips move as pitched 2-D copies, which are DMA transfers from
  // pinned memory but row-by-row staging from pageable memory (8192^3: 15.0 ms in one harness, 23 ms in another, against
  // 15.7 ms for the row-panel form -- profiles/r02/host_pipeline_v2.jsonl, configs_v19.jsonl)
  auto pinned = [](const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
      (void)hipGetLastError();  // an ordinary pageable pointer is "invalid value" to the runtime: not an error here
      return false;
    }
    return at.type == hipMemoryTypeHost;
  };
  if (!fused && g_ctx.host_pipeline_2d && pinned(B) && pinned(C) && csA == 1 && csB == 1 && csC == 1 && rsA >= K && rsB >= N && rsC >= N && M >= 2048 &&
      N >= 2048 && bn * sizeof(T) >= ((size_t)32 << 20) && (an + bn + cn) * sizeof(T) >= ((size_t)128 << 20))
    return gemm_host_pipelined2d<T>(M, N, K, alpha, A, rsA, B, rsB, beta, C, rsC, dA0, dB0, dC0, c_up);
  if (!fused && panels_disjoint && M >= 2048 && (an + bn + cn) * sizeof(T) >= ((size_t)64 << 20))
    return gemm_host_pipelined<T>(M, N, K, alpha, A, rsA, csA, B + blo, bn, rsB, csB, beta, C, rsC, csC, dA0, dB0,
                                  (T *)dB, dC0, c_up);

  HIP_TRY(hipMemcpy(dA, A + alo, an * sizeof(T), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dB, B + blo, bn * sizeof(T), hipMemcpyHostToDevice));
  if (c_up) HIP_TRY(hipMemcpy(dC, C + clo, cn * sizeof(T), hipMemcpyHostToDevice));
  GemmArgs<T> a = make_args<T>(1, M, N, K, alpha, dA0, rsA, csA, 0, dB0, rsB, csB, 0, beta, dC0, rsC, csC, 0);
  if (fused) epi_apply(a, &depi);
  HIP_TRY(run_gemm<T>(a, nullptr));
  HIP_TRY(hipMemcpy(C + clo, dC, cn * sizeof(T), hipMemcpyDeviceToHost));  // synchronises
  return LASER_HIP_OK;
}

// ---- pre-pack ------------------------------------------------------------------------------------
// Panel image = dense row-major copy, zero-padded so every tile configuration is "full":
// rows of A / cols of B to a multiple of 256, k to a multiple of 32.
constexpr int64_t kPadMN = 256, kPadK = 32;
inline int64_t rup(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Host pre-pack buffers are SELF-CONTAINED, like the reference's (gemm_prepacked.nim:111-135: the packed panels live in the caller's
// `mem_required` bytes; Design.md:5-7: Laser keeps no memory of its own): a 64-byte header followed by the tile-padded panel image.
// The device copy gemm_packed multiplies from is a CACHE of that image, keyed by the header's id (unique per prepack call) and the
// device: made by the prepack call itself, re-made from the caller's buffer whenever it is missing (evicted, made on another device,
// the buffer is a memcpy of the original), dropped by laser_hip_gemm_prepack_release / finalize, and bounded (least recently used
// first) -- a caller who simply frees its buffers, as a Laser caller would, cannot leak HBM (VERDICT r4 missing #4).
constexpr uint64_t kMagic = 0x4c41534552484951ull;  // "LASERHIQ": layout 2 (header + image)
struct PackHandle {  // the first 64 bytes of the caller's (64-B aligned) host buffer
  uint64_t magic, id;
  int64_t M, N, K;
  int32_t is_a, elem;
  uint64_t image_bytes;
  uint64_t sum;        // fingerprint of the image (samples + length): a device copy is only ever used for the image it was made from
};
static_assert(sizeof(PackHandle) <= 64, "handle must fit the alignment unit");
constexpr size_t kPackHeader = 64;
struct DevPanel {
  void *ptr = nullptr;
  size_t bytes = 0;
  int device = 0;
  int pins = 0;            // calls multiplying from it right now: never evicted
  uint64_t last_use = 0;
  uint64_t id = 0, sum = 0;   // of the header it was made for (release drops by id; a hit is checked against both)
};
std::unordered_map<uint64_t, DevPanel> g_panels;      // key = mix(id, fingerprint, shape) with the device ordinal in the low byte
size_t g_panel_bytes = 0;
uint64_t g_panel_clock = 0;
constexpr size_t kPanelCacheMax = (size_t)16 << 30;
inline uint64_t mix64(uint64_t x) {      // splitmix64 finaliser
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
// Ids are unique ACROSS processes, not a counter from 1: a buffer packed in one process and handed to another -- or to
// a forked child -- must not meet a cached panel that the other process made under the same number.  A per-process random salt
// (re-drawn in a forked child) goes through a 64-bit mixer with the counter; the cache key also carries the header's shape and
// the image fingerprint, so even an id collision cannot pair a header with another image's device copy.
uint64_t g_id_salt = 0;
uint64_t g_next_id = 1;
void draw_id_salt() {
  uint64_t s = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32) ^ (uint64_t)(uintptr_t)&g_id_salt;
  if (FILE *f = fopen("/dev/urandom", "rb")) {
    uint64_t r = 0;
    if (fread(&r, sizeof r, 1, f) == 1) s ^= r;
    fclose(f);
  }
  g_id_salt = mix64(s);
}
uint64_t fresh_id_locked() {
  static bool once = [] {
    draw_id_salt();
    (void)pthread_atfork(nullptr, nullptr, [] { draw_id_salt(); });
    return true;
  }();
  (void)once;
  return mix64(g_id_salt ^ mix64(g_next_id++));
}
// fingerprint of a panel image in host memory: its length and 256 samples of 8 bytes spread over it (cheap beside the copies a
// pre-pack makes; it guards the cache against headers that name the same id for different images, not against an adversary)
uint64_t image_fingerprint(const void *img, size_t bytes) {
  uint64_t h = mix64(bytes);
  const size_t words = bytes / 8;
  if (!words) return h;
  const size_t step = std::max<size_t>(1, words / 256);
  for (size_t i = 0; i < words; i += step) {
    uint64_t w;
    memcpy(&w, (const char *)img + 8 * i, 8);
    h = mix64(h ^ w);
  }
  uint64_t w;
  memcpy(&w, (const char *)img + 8 * (words - 1), 8);
  return mix64(h ^ w);
}
inline uint64_t panel_key(const PackHandle &h, int dev) {
  uint64_t k = mix64(h.id ^ mix64(h.sum ^ mix64((uint64_t)h.M * 0x100000001b3ull ^ (uint64_t)h.N * 0x1000193ull ^ (uint64_t)h.K ^ ((uint64_t)h.is_a << 40) ^ ((uint64_t)h.elem << 48))));
  return (k << 8) | (uint64_t)(dev & 0xff);
}
// (g_mu held) make room for `need` more bytes: least recently used unpinned panels go first
void panel_cache_evict_locked(size_t need) {
  while (g_panel_bytes + need > kPanelCacheMax) {
    auto victim = g_panels.end();
    for (auto it = g_panels.begin(); it != g_panels.end(); ++it)
      if (it->second.pins == 0 && (victim == g_panels.end() || it->second.last_use < victim->second.last_use)) victim = it;
    if (victim == g_panels.end()) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (victim->second.device != cur) (void)hipSetDevice(victim->second.device);
    (void)hipFree(victim->second.ptr);      // (waits for the device: no launch still reads it)
    if (victim->second.device != cur) (void)hipSetDevice(cur);
    g_panel_bytes -= victim->second.bytes;
    g_panels.erase(victim);
  }
}
// (g_mu held) every device's copy of panel `id`, unless a call is multiplying from one
void panel_cache_drop_locked(uint64_t id) {
  for (auto it = g_panels.begin(); it != g_panels.end();) {
    if (it->second.id == id && it->second.pins == 0) {
      int cur = 0;
      (void)hipGetDevice(&cur);
      if (it->second.device != cur) (void)hipSetDevice(it->second.device);
      (void)hipFree(it->second.ptr);
      if (it->second.device != cur) (void)hipSetDevice(cur);
      g_panel_bytes -= it->second.bytes;
      it = g_panels.erase(it);
    } else {
      ++it;
    }
  }
}

// (g_mu held) a panel allocation that gives cached panels back before it gives up: out of memory -> every unpinned panel of this
// device is evicted (least recently used first, all of them if need be) and the allocation is retried once (a caller who
// frees buffers without `release` may leave the bounded cache holding most of a GPU's memory)
int panel_alloc_locked(void **ptr, size_t bytes, int dev) {
  hipError_t e = hipMalloc(ptr, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    bool freed = false;
    for (auto it = g_panels.begin(); it != g_panels.end();) {
      if (it->second.device == dev && it->second.pins == 0) {
        (void)hipFree(it->second.ptr);
        g_panel_bytes -= it->second.bytes;
        it = g_panels.erase(it);
        freed = true;
      } else {
        ++it;
      }
    }
    if (freed) e = hipMalloc(ptr, bytes);
  }
  if (e != hipSuccess) {
    *ptr = nullptr;
    return fail(LASER_HIP_E_HIP, "allocating a pre-packed panel (%zu bytes): %s", bytes, hipGetErrorString(e));
  }
  return LASER_HIP_OK;
}
int panel_alloc(void **ptr, size_t bytes, int dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  return panel_alloc_locked(ptr, bytes, dev);
}

// device tensor storage: live blocks (ptr -> rounded size) and the free list keyed by size
constexpr size_t kStorageCacheMax = (size_t)32 << 30;
// live blocks: ptr -> key; free list keyed by the same key = rounded size | device ordinal << 56 (a block is only
// ever handed back to a caller on the device it was allocated on)
std::unordered_map<void *, size_t> g_live_storage;
std::unordered_map<size_t, std::vector<void *>> g_free_storage;
inline size_t storage_key(size_t rounded, int dev) { return rounded | ((size_t)(dev & 0xff) << 56); }
inline size_t storage_size(size_t key) { return key & (((size_t)1 << 56) - 1); }
size_t g_free_storage_bytes = 0;
void storage_trim() {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &kv : g_free_storage)
    for (void *p : kv.second) {
      (void)hipFree(p);
      g_live_storage.erase(p);
    }
  g_free_storage.clear();
  g_free_storage_bytes = 0;
}
template <typename T>
int64_t prepack_bytes(bool is_a, int64_t M, int64_t N, int64_t K) {
  if (M < 0 || N < 0 || K < 0) return 0;
  const int64_t x = is_a ? rup(M, kPadMN) : rup(N, kPadMN);
  const int64_t b = (int64_t)sizeof(T) * x * rup(K, kPadK);
  return std::max<int64_t>(b, 64) + (int64_t)kPackHeader;      // (host form: header + image; device form: the image alone, at offset 0)
}
template <typename T>
int64_t prepack_image_bytes(bool is_a, int64_t M, int64_t N, int64_t K) {
  return prepack_bytes<T>(is_a, M, N, K) - (int64_t)kPackHeader;
}

template <typename T>
int prepack_dev(bool is_a, void *d_dst, int64_t M, int64_t N, int64_t K, const T *src, int64_t rs, int64_t cs,
                void *stream) {
  if (int rc = ensure_init()) return rc;
  if (!d_dst || !src) return fail(LASER_HIP_E_INVALID, "null pointer");
  if (is_a)  // A image: [Mpad][Kpad], k contiguous
    HIP_TRY(launch_pack_pad<T>((T *)d_dst, rup(M, kPadMN), rup(K, kPadK), src, M, K, rs, cs, (hipStream_t)stream));
  else       // B image: [Kpad][Npad], n contiguous
    HIP_TRY(launch_pack_pad<T>((T *)d_dst, rup(K, kPadK), rup(N, kPadMN), src, K, N, rs, cs, (hipStream_t)stream));
  return LASER_HIP_OK;
}

template <typename T>
int prepack_host(bool is_a, void *dst, int64_t M, int64_t N, int64_t K, const T *src, int64_t rs, int64_t cs) {
  if (!dst || !src) return fail(LASER_HIP_E_INVALID, "null pointer");
  // same precondition as the reference's doAssert (gemm_prepacked.nim:125, :208)
  if ((reinterpret_cast<uintptr_t>(dst) & 63) != 0)
    return fail(LASER_HIP_E_INVALID, "The destination pointer must be 64-byte aligned");
  if (M < 0 || N < 0 || K < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = ensure_init()) return rc;
  HostCall hc;
  if (hc.rc) return hc.rc;
  const int64_t R = is_a ? M : K, Cc = is_a ? K : N;
  int64_t lo, hi;
  view_span(std::max<int64_t>(R, 1), std::max<int64_t>(Cc, 1), rs, cs, &lo, &hi);
  const size_t n = (size_t)(hi - lo + 1);
  void *dsrc;
  if (int rc = scratch_get(3, n * sizeof(T), &dsrc)) return rc;
  if (R > 0 && Cc > 0) HIP_TRY(hipMemcpy(dsrc, src + lo, n * sizeof(T), hipMemcpyHostToDevice));
  DevPanel p;
  p.bytes = (size_t)prepack_image_bytes<T>(is_a, M, N, K);
  (void)hipGetDevice(&p.device);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    panel_cache_evict_locked(p.bytes);
  }
  if (int rc = panel_alloc(&p.ptr, p.bytes, p.device)) return rc;
  hipError_t e = is_a ? launch_pack_pad<T>((T *)p.ptr, rup(M, kPadMN), rup(K, kPadK), (const T *)dsrc - lo, M, K, rs, cs, nullptr)
                      : launch_pack_pad<T>((T *)p.ptr, rup(K, kPadK), rup(N, kPadMN), (const T *)dsrc - lo, K, N, rs, cs, nullptr);
  // the image goes into the caller's buffer (what makes the buffer self-contained); the device copy stays as the cache's first entry
  if (e == hipSuccess) e = hipMemcpy((char *)dst + kPackHeader, p.ptr, p.bytes, hipMemcpyDeviceToHost);   // synchronises
  if (e != hipSuccess) {
    (void)hipFree(p.ptr);
    return fail(LASER_HIP_E_HIP, "pre-pack: %s", hipGetErrorString(e));
  }
  PackHandle h;
  memset(&h, 0, sizeof h);
  h.magic = kMagic;
  h.M = M; h.N = N; h.K = K;
  h.is_a = is_a ? 1 : 0;
  h.elem = (int32_t)sizeof(T);
  h.image_bytes = p.bytes;
  h.sum = image_fingerprint((const char *)dst + kPackHeader, p.bytes);
  std::lock_guard<std::mutex> lk(g_mu);  // the panel cache
  h.id = fresh_id_locked();
  // re-packing into a buffer that still holds a live header drops the old image's device copies first
  PackHandle old;
  memcpy(&old, dst, sizeof old);
  if (old.magic == kMagic) panel_cache_drop_locked(old.id);
  memcpy(dst, &h, sizeof h);
  p.last_use = ++g_panel_clock;
  p.id = h.id; p.sum = h.sum;
  g_panels[panel_key(h, p.device)] = p;
  g_panel_bytes += p.bytes;
  return LASER_HIP_OK;
}

// (g_mu held) the device copy of a host pre-pack buffer on the current device, PINNED (panel_unpin when the product is done); made
// from the caller's buffer when the cache does not hold it
int resolve_handle(const void *packed, bool want_a, int elem, int64_t M, int64_t N, int64_t K, void **dptr, uint64_t *key_out) {
  if (!packed) return fail(LASER_HIP_E_INVALID, "null packed buffer");
  PackHandle h;
  memcpy(&h, packed, sizeof h);
  if (h.magic != kMagic) return fail(LASER_HIP_E_HANDLE, "buffer does not hold a pre-packed operand (never packed, or released)");
  if ((h.is_a != 0) != want_a || h.elem != elem)
    return fail(LASER_HIP_E_HANDLE, "pre-packed buffer is for another operand / element type");
  if (want_a ? (h.M != M || h.K != K) : (h.N != N || h.K != K))
    return fail(LASER_HIP_E_HANDLE, "pre-packed buffer was made for a different shape");
  // (the header is caller memory: the image size it states must be the one this shape has before it sizes an allocation and a copy)
  const int64_t want_bytes = std::max<int64_t>((int64_t)elem * rup(want_a ? M : N, kPadMN) * rup(K, kPadK), 64);
  if ((int64_t)h.image_bytes != want_bytes) return fail(LASER_HIP_E_HANDLE, "pre-packed buffer header is corrupt (image size)");
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t key = panel_key(h, dev);
  auto it = g_panels.find(key);
  // A hit is used only if it is the copy of THIS image: same id, same fingerprint, same size (the key mixes all of them, so anything
  // else is a 64-bit hash collision -- or a header someone edited).  Such an entry is left alone (a call may be multiplying from it)
  // and the operand is refused rather than multiplied from the wrong matrix / read past a smaller panel.
  if (it != g_panels.end() && (it->second.id != h.id || it->second.sum != h.sum || it->second.bytes != (size_t)h.image_bytes))
    return fail(LASER_HIP_E_HANDLE, "pre-packed buffer header does not match the cached device image (foreign or edited header)");
  if (it == g_panels.end()) {      // evicted / another device / a copy of the buffer in a process that never packed it: upload the image
    if (image_fingerprint((const char *)packed + kPackHeader, (size_t)h.image_bytes) != h.sum)
      return fail(LASER_HIP_E_HANDLE, "pre-packed buffer does not hold the image its header describes");
    DevPanel p;
    p.bytes = (size_t)h.image_bytes;
    p.device = dev;
    p.id = h.id; p.sum = h.sum;
    panel_cache_evict_locked(p.bytes);
    if (int rc = panel_alloc_locked(&p.ptr, p.bytes, dev)) return rc;
    const hipError_t e = hipMemcpy(p.ptr, (const char *)packed + kPackHeader, p.bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(p.ptr);
      return fail(LASER_HIP_E_HIP, "uploading a pre-packed operand: %s", hipGetErrorString(e));
    }
    g_panel_bytes += p.bytes;
    it = g_panels.emplace(key, p).first;
  }
  it->second.pins++;
  it->second.last_use = ++g_panel_clock;
  *dptr = it->second.ptr;
  *key_out = key;
  return LASER_HIP_OK;
}
void panel_unpin(uint64_t key) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_panels.find(key);
  if (it != g_panels.end() && it->second.pins > 0) it->second.pins--;
}

template <typename T>
GemmArgs<T> packed_args(int64_t M, int64_t N, int64_t K, T alpha, const void *dA, const void *dB, T beta, T *dC,
                        int64_t rsC, int64_t csC) {
  GemmArgs<T> a = make_args<T>(1, M, N, K, alpha, (const T *)dA, rup(K, kPadK), 1, 0, (const T *)dB,
                               rup(N, kPadMN), 1, 0, beta, dC, rsC, csC, 0);
  a.Mext = rup(M, kPadMN);
  a.Next = rup(N, kPadMN);
  a.Kext = rup(K, kPadK);
  return a;
}

template <typename T>
int packed_dev(int64_t M, int64_t N, int64_t K, T alpha, const void *dA, const void *dB, T beta, T *dC,
               int64_t rsC, int64_t csC, void *stream) {
  if (M < 0 || N < 0 || K < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = ensure_init()) return rc;
  if (M == 0 || N == 0 || K == 0) return LASER_HIP_OK;
  if (!dA || !dB || !dC) return fail(LASER_HIP_E_INVALID, "null pointer");
  HIP_TRY(run_gemm<T>(packed_args<T>(M, N, K, alpha, dA, dB, beta, dC, rsC, csC), (hipStream_t)stream));
  return LASER_HIP_OK;
}

template <typename T>
int packed_host(int64_t M, int64_t N, int64_t K, T alpha, const void *pA, const void *pB, T beta, T *C,
                int64_t rsC, int64_t csC) {
  if (M < 0 || N < 0 || K < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = ensure_init()) return rc;
  if (M == 0 || N == 0 || K == 0) return LASER_HIP_OK;
  if (!C) return fail(LASER_HIP_E_INVALID, "null pointer");
  HostCall hc;
  if (hc.rc) return hc.rc;
  void *dA, *dB, *dC;
  uint64_t keyA = 0, keyB = 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);  // the panel cache
    if (int rc = resolve_handle(pA, true, (int)sizeof(T), M, N, K, &dA, &keyA)) return rc;
    if (int rc = resolve_handle(pB, false, (int)sizeof(T), M, N, K, &dB, &keyB)) {
      g_panels[keyA].pins--;
      return rc;
    }
  }
  struct Unpin {      // both panels stay in the cache until this call's product has finished (the D2H copy below synchronises)
    uint64_t a, b;
    ~Unpin() { panel_unpin(a); panel_unpin(b); }
  } unpin{keyA, keyB};
  int64_t clo, chi;
  view_span(M, N, rsC, csC, &clo, &chi);
  const size_t cn = (size_t)(chi - clo + 1);
  if (int rc = scratch_get(2, cn * sizeof(T), &dC)) return rc;
  if (beta != (T)0 || cn != (size_t)M * (size_t)N)
    HIP_TRY(hipMemcpy(dC, C + clo, cn * sizeof(T), hipMemcpyHostToDevice));
  HIP_TRY(run_gemm<T>(packed_args<T>(M, N, K, alpha, dA, dB, beta, (T *)dC - clo, rsC, csC), nullptr));
  HIP_TRY(hipMemcpy(C + clo, dC, cn * sizeof(T), hipMemcpyDeviceToHost));
  return LASER_HIP_OK;
}

// ---- transposes ------------------------------------------------------------------------------------
int transpose_host(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC, int elem) {
  if (N < 0 || NR < 0 || NC < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = ensure_init()) return rc;
  const size_t bytes = (size_t)N * NR * NC * elem;
  if (bytes == 0) return LASER_HIP_OK;
  if (!dst || !src) return fail(LASER_HIP_E_INVALID, "null pointer");
  HostCall hc;
  if (hc.rc) return hc.rc;
  void *ds, *dd;
  if (int rc = scratch_get(0, bytes, &ds)) return rc;
  if (int rc = scratch_get(2, bytes, &dd)) return rc;
  HIP_TRY(hipMemcpy(ds, src, bytes, hipMemcpyHostToDevice));
  HIP_TRY(launch_transpose_batched(dd, ds, N, NR, NC, elem, nullptr));
  HIP_TRY(hipMemcpy(dst, dd, bytes, hipMemcpyDeviceToHost));
  return LASER_HIP_OK;
}

// ---- convolution -------------------------------------------------------------------------------------
void out_hw(int64_t iH, int64_t iW, int64_t kH, int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW,
            int64_t *oH, int64_t *oW) {
  // conv2d_common.nim:43-44 with dilation 1
  *oH = 1 + (iH + 2 * pH - kH) / sH;
  *oW = 1 + (iW + 2 * pW - kW) / sW;
}

int conv_check(int64_t iN, int64_t iC, int64_t iH, int64_t iW, int64_t c_out, int64_t c_in, int64_t kH, int64_t kW,
               int64_t pH, int64_t pW, int64_t sH, int64_t sW) {
  if (iN < 0 || iC <= 0 || iH <= 0 || iW <= 0 || c_out <= 0 || kH <= 0 || kW <= 0 || pH < 0 || pW < 0)
    return fail(LASER_HIP_E_INVALID, "bad convolution shape");
  if (c_in != iC) return fail(LASER_HIP_E_INVALID, "kernel c_in (%lld) != input channels (%lld)", (long long)c_in, (long long)iC);
  // conv2d_common.nim:33-34: doAssert 0 < sH and sH < iH (same for W)
  if (!(0 < sH && sH < iH && 0 < sW && sW < iW)) return fail(LASER_HIP_E_INVALID, "strides must satisfy 0 < s < input size");
  if (iH + 2 * pH < kH || iW + 2 * pW < kW) return fail(LASER_HIP_E_INVALID, "kernel larger than padded input");
  return LASER_HIP_OK;
}

// The implicit-GEMM loader keeps 8-bit row/column validity bitmaps and 32-bit in-image offsets:
// kernels up to 8x8 and images below 2^30 elements; anything else goes through the explicit workspace.
bool conv_takes_implicit(int64_t iC, int64_t iH, int64_t iW, int64_t kH, int64_t kW, int64_t pH, int64_t pW,
                         int64_t sH, int64_t sW) {
  int64_t oH, oW;
  out_hw(iH, iW, kH, kW, pH, pW, sH, sW, &oH, &oW);
  const int64_t reach = (oH * sH + kH + pH) * iW + oW * sW + kW + pW;  // largest |window origin| offset
  return g_ctx.conv_implicit && kH <= 8 && kW <= 8 && iC * iH * iW + reach < (1ll << 30) && iC * kH * kW < (1ll << 30);
}

std::atomic<int> g_conv_1x1_implicit{0};   // option "conv_1x1_implicit" (probes): no GEMM shortcut for 1x1 / stride 1 / pad 0

int conv_dev(float *dout, const float *din, int64_t iN, int64_t iC, int64_t iH, int64_t iW, const float *dker,
             int64_t c_out, int64_t kH, int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *dws,
             hipStream_t s, const float *dbias = nullptr, int act = 0) {
  // per-output-channel bias = one value per row of the [C_out x oH*oW] product, shared by all images
  Epi<float> epi;
  epi.bias = dbias; epi.rs = 1; epi.cs = 0; epi.bs = 0; epi.act = act;
  if (int rc = epi_check<float>(&epi)) return rc;
  int64_t oH, oW;
  out_hw(iH, iW, kH, kW, pH, pW, sH, sW, &oH, &oW);
  const int64_t M = c_out, K = iC * kH * kW, N = oH * oW;
  // 1x1 shortcut (conv2d_im2col.nim:121,128,151-153): the input already is the [K, N] matrix --
  // taken only for stride 1 / no padding, where that is actually true.
  // (option conv_1x1_implicit = 1, probes: the 1x1 / stride 1 / no padding case through the implicit-GEMM kernels as well)
  const bool direct = (kH * kW == 1) && pH == 0 && pW == 0 && sH == 1 && sW == 1 && !g_conv_1x1_implicit;
  if (!direct && conv_takes_implicit(iC, iH, iW, kH, kW, pH, pW, sH, sW)) {
    // implicit GEMM: im2col's index arithmetic runs inside the B-tile loader, nothing is materialised
    GemmArgs<float> a = make_args<float>(iN, M, N, K, 1.0f, dker, K, 1, 0, din, 0, 1, iC * iH * iW, 0.0f, dout, N, 1, M * N);
    a.cH = (int32_t)iH; a.cW = (int32_t)iW; a.ckH = (int32_t)kH; a.ckW = (int32_t)kW; a.coW = (int32_t)oW;
    a.cpH = (int32_t)pH; a.cpW = (int32_t)pW; a.csH = (int32_t)sH; a.csW = (int32_t)sW;
    epi_apply(a, &epi);
    HIP_TRY(launch_conv_implicit_f32(a, f32_cfg_now(), g_ctx.float_mode == LASER_HIP_F32_LASER_ORDER, s));
    return LASER_HIP_OK;
  }
  const float *Bm = din;
  int64_t bsB = iC * iH * iW;
  if (!direct) {
    if (!dws) return fail(LASER_HIP_E_INVALID, "explicit im2col path needs a workspace");
    HIP_TRY(launch_im2col_f32(dws, oH, oW, din, iN, iC, iH, iW, kH, kW, pH, pW, sH, sW, s));
    Bm = dws;
    bsB = K * N;
  }
  // O[n] (M x N) = F (M x K) . W[n] (K x N), alpha = 1, beta = 0 (conv2d_im2col.nim:104-105,161-166);
  // all images in ONE batched launch: A = filter shared by every image (batch stride 0)
  GemmArgs<float> a = make_args<float>(iN, M, N, K, 1.0f, dker, K, 1, 0, Bm, N, 1, bsB, 0.0f, dout, N, 1, M * N);
  epi_apply(a, &epi);
  HIP_TRY(run_gemm<float>(a, s));
  return LASER_HIP_OK;
}

// cblas enums: benchmarks/third_party/blas.nim:12-16
template <typename T>
int cblas_gemm(int order, int tA, int tB, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t lda,
                      const T *B, int64_t ldb, T beta, T *C, int64_t ldc) {
  if ((order != 101 && order != 102) || tA < 111 || tA > 113 || tB < 111 || tB > 113)
    return fail(LASER_HIP_E_INVALID, "bad cblas order/transpose enum");
  const bool rowm = order == 101;
  // op(X)[r,c]: row-major no-trans -> (ld,1); row-major trans -> (1,ld); col-major swaps them
  auto strides = [&](bool trans, int64_t ld, int64_t *rs, int64_t *cs) {
    const bool r_contig_c = rowm != trans;  // element [r, c+1] adjacent
    *rs = r_contig_c ? ld : 1;
    *cs = r_contig_c ? 1 : ld;
  };
  int64_t rsA, csA, rsB, csB;
  strides(tA != 111, lda, &rsA, &csA);
  strides(tB != 111, ldb, &rsB, &csB);
  const int64_t rsC = rowm ? ldc : 1, csC = rowm ? 1 : ldc;
  return gemm_host<T>(M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC);
}


}  // namespace

// =====================================================================================================
namespace {
template <typename U>
int copy_strided_api(void *dst, const int64_t *ds, const void *src, const int64_t *ss, const int64_t *shape,
                            int rank, void *stream) {
  if (rank < 0 || rank > kMaxRank) return fail(LASER_HIP_E_INVALID, "rank %d outside 0..%d (LASER_MAXRANK)", rank, kMaxRank);
  if (rank > 0 && (!ds || !ss || !shape)) return fail(LASER_HIP_E_INVALID, "null shape/strides");
  int64_t total = 1;
  for (int d = 0; d < rank; d++) {
    if (shape[d] < 0) return fail(LASER_HIP_E_INVALID, "negative extent");
    total *= shape[d];
  }
  if (int rc = ensure_init()) return rc;
  if (total == 0) return LASER_HIP_OK;
  if (!dst || !src) return fail(LASER_HIP_E_INVALID, "null buffer");
  // both sides C-contiguous: a plain device-to-device copy
  bool contiguous = true;
  int64_t run = 1;
  for (int d = rank - 1; d >= 0; d--) {
    if (shape[d] != 1 && (ds[d] != run || ss[d] != run)) contiguous = false;
    run *= shape[d];
  }
  if (contiguous) {
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)total * sizeof(U), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return LASER_HIP_OK;
  }
  HIP_TRY(launch_copy_strided<U>((U *)dst, ds, (const U *)src, ss, shape, rank, (hipStream_t)stream));
  return LASER_HIP_OK;
}
}  // namespace

namespace laser_hip {
int api_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int api_ensure_init() { return ensure_init(); }
void api_set_thread_device(int device) { tl_device = device; }
void api_set_thread_f32_config(int cfg) { tl_f32_cfg = cfg; }
void api_set_thread_asm_tile(int tile_class) { asm_set_thread_tile(tile_class); }
int api_thread_device() { return tl_device; }
}  // namespace laser_hip

extern "C" {

int laser_hip_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  return ensure_init_locked(device);
}

int laser_hip_finalize(void) {
  // lock order everywhere: a device's mutex BEFORE g_mu (the host-pointer entry points hold their device's mutex when
  // they touch the registries) -- so the per-device state is released first, one device at a time, then the registries
  for (DeviceCtx &D : g_dev) {
    std::lock_guard<std::mutex> dl(D.mu);
    if (D.device < 0) continue;
    (void)hipSetDevice(D.device);
    for (int i = 0; i < 6; i++) {
      if (D.scratch[i]) (void)hipFree(D.scratch[i]);
      D.scratch[i] = nullptr;
      D.scratch_sz[i] = 0;
    }
    if (D.s_up) {
      (void)hipStreamDestroy(D.s_up);
      (void)hipStreamDestroy(D.s_comp);
      (void)hipStreamDestroy(D.s_down);
      D.s_up = D.s_comp = D.s_down = nullptr;
    }
    if (D.zc) (void)hipHostFree(D.zc);
    D.zc = nullptr;
    D.zc_sz = 0;
    D.device = -1;
  }
  asm_kernels_release();
  scratch_pools_trim();
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx.device >= 0) (void)hipSetDevice(g_ctx.device);
  for (auto &kv : g_panels) {
    (void)hipSetDevice(kv.second.device);
    (void)hipFree(kv.second.ptr);
  }
  g_panels.clear();
  g_panel_bytes = 0;
  if (g_ctx.device >= 0) (void)hipSetDevice(g_ctx.device);
  for (auto &kv : g_free_storage)
    for (void *p : kv.second) {
      (void)hipFree(p);
      g_live_storage.erase(p);
    }
  g_free_storage.clear();
  g_free_storage_bytes = 0;
  g_ctx.ready = false;
  return LASER_HIP_OK;
}

const char *laser_hip_last_error(void) { return g_err.c_str(); }
const char *laser_hip_version(void) { return "laser_hip 0.2.0 (gfx950)"; }
int laser_hip_abi_version(void) { return LASER_HIP_ABI_VERSION; }
int laser_hip_plan_f32(int64_t M, int64_t N, int64_t K, int laser_order, int cus, int64_t *out8) {
  if (!out8) return fail(LASER_HIP_E_INVALID, "null pointer");
  if (asm_plan_f32(M, N, K, laser_order, cus, out8)) return fail(LASER_HIP_E_INVALID, "laser_hip_plan_f32: bad shape or CU count");
  return LASER_HIP_OK;
}
int laser_hip_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
const char *laser_hip_arch(void) {
  if (ensure_init()) return "";
  return g_ctx.arch.c_str();
}

int laser_hip_set_float_mode(int mode) {
  if (mode != LASER_HIP_F32_LASER_ORDER && mode != LASER_HIP_F32_FAST) return fail(LASER_HIP_E_INVALID, "bad float mode");
  g_ctx.float_mode = mode;
  return LASER_HIP_OK;
}
int laser_hip_get_float_mode(void) { return g_ctx.float_mode; }
int laser_hip_set_f32_config(int cfg) {
  if (cfg >= gemm_f32_config_count()) return fail(LASER_HIP_E_INVALID, "bad f32 config index");
  g_ctx.f32_cfg = cfg < 0 ? -1 : cfg;
  return LASER_HIP_OK;
}
int laser_hip_f32_config_count(void) { return gemm_f32_config_count(); }
// ---- options: every tuning / A-B switch behind ONE entry point (name, value); diagnostics behind laser_hip_get_option ----
// (the header documents each name; unknown names are an error, never silently ignored)
int laser_hip_set_option(const char *name, int value) {
  if (!name) return fail(LASER_HIP_E_INVALID, "set_option: null name");
  const std::string n(name);
  const bool on = value != 0;
  if (n == "f32_asm") g_f32_asm = value < 0 ? 0 : value > 2 ? 2 : value;
  else if (n == "f64_asm") g_f64_asm = value < 0 ? 0 : value > 2 ? 2 : value;
  else if (n == "i32_asm") g_i32_asm = value < 0 ? 0 : value > 2 ? 2 : value;
  else if (n == "int_group_m") g_int_group_m = value < 1 ? 1 : value > 64 ? 64 : value;
  else if (n == "f64_mfma") g_ctx.f64_mfma = on;
  else if (n == "i32_mfma") g_ctx.i32_mfma = on;
  else if (n == "i64_mfma") g_ctx.i64_mfma = on;
  else if (n == "conv_implicit") g_ctx.conv_implicit = on;
  else if (n == "conv_patch") g_conv_patch = on;
  else if (n == "conv_direct") g_conv_direct = value < 0 ? 0 : value > 3 ? 3 : value;
  else if (n == "conv_kslice") g_conv_kslice = on;
  else if (n == "conv_tail") g_conv_tail = on;
  else if (n == "conv_cut_always") g_conv_cut_always = on;
  else if (n == "conv_1x1_implicit") g_conv_1x1_implicit = on;
  else if (n == "conv_walk") g_conv_walk = value < 0 ? 0 : value > 65535 ? 65535 : value;
  else if (n == "host_pipeline_2d") g_ctx.host_pipeline_2d = on;
  else if (n == "zero_copy_poll") g_ctx.zc_poll = on;
  else if (n == "skinny") g_ctx.skinny = on;
  else if (n == "small_path") g_small_path = on;
  else if (n == "split_tail") g_split_tail = on;
  else if (n == "asm_plan") g_asm_plan = value < 0 ? 0 : value > 4 ? 4 : value;
  else if (n == "asm_kernel") g_asm_kernel = value < 0 ? -1 : value;
  else if (n == "asm_tile") g_asm_tile = value < 0 || value > 9 ? -1 : value;
  else if (n == "thread_asm_tile") asm_set_thread_tile(value < -1 || value > 9 ? -2 : value);
  else if (n == "im2col_band") g_im2col_band = value < 0 ? 0 : value;
  else if (n == "asm_wgs") g_asm_wgs = value < 0 ? 0 : value;
  else if (n == "asm_slice") g_asm_slice = value < 0 ? 0 : value;
  else if (n == "asm_noseed") g_asm_noseed = on;
  else if (n == "asm_test_giveup") g_asm_giveup = on;
  else if (n == "asm_group_m") g_asm_group_m = value < 0 ? 0 : value;
  else if (n == "slice_parallel") g_ctx.slice_parallel = on;
  else if (n == "slice_parallel_min") g_ctx.slice_parallel_min = value < 2 ? 2 : value;
  else if (n == "slice_parallel_tiles") g_ctx.slice_parallel_tiles = value < 0 ? 0 : value;
  else return fail(LASER_HIP_E_INVALID, "set_option: unknown option '%s'", name);
  return LASER_HIP_OK;
}
int laser_hip_get_option(const char *name, int64_t *value) {
  if (!name || !value) return fail(LASER_HIP_E_INVALID, "get_option: null argument");
  const std::string n(name);
  if (n == "f32_asm") *value = g_f32_asm;
  else if (n == "f64_asm") *value = g_f64_asm;
  else if (n == "last_f64_asm") *value = g_last_f64_asm;
  else if (n == "i32_asm") *value = g_i32_asm;
  else if (n == "int_group_m") *value = g_int_group_m;
  else if (n == "last_i32_asm") *value = g_last_i32_asm;
  else if (n == "f64_mfma") *value = g_ctx.f64_mfma;
  else if (n == "i32_mfma") *value = g_ctx.i32_mfma;
  else if (n == "i64_mfma") *value = g_ctx.i64_mfma;
  else if (n == "conv_implicit") *value = g_ctx.conv_implicit;
  else if (n == "conv_patch") *value = g_conv_patch;
  else if (n == "conv_direct") *value = g_conv_direct;
  else if (n == "conv_kslice") *value = g_conv_kslice;
  else if (n == "conv_tail") *value = g_conv_tail;
  else if (n == "conv_cut_always") *value = g_conv_cut_always;
  else if (n == "conv_1x1_implicit") *value = g_conv_1x1_implicit;
  else if (n == "conv_walk") *value = g_conv_walk;
  else if (n == "host_pipeline_2d") *value = g_ctx.host_pipeline_2d;
  else if (n == "zero_copy_poll") *value = g_ctx.zc_poll;
  else if (n == "skinny") *value = g_ctx.skinny;
  else if (n == "small_path") *value = g_small_path;
  else if (n == "split_tail") *value = g_split_tail;
  else if (n == "asm_plan") *value = g_asm_plan;
  else if (n == "asm_kernel") *value = g_asm_kernel;
  else if (n == "asm_tile") *value = g_asm_tile;
  else if (n == "thread_asm_tile") *value = asm_get_thread_tile();
  else if (n == "im2col_band") *value = g_im2col_band;
  else if (n == "asm_wgs") *value = g_asm_wgs;
  else if (n == "asm_slice") *value = g_asm_slice;
  else if (n == "asm_noseed") *value = g_asm_noseed;
  else if (n == "asm_test_giveup") *value = g_asm_giveup;
  else if (n == "asm_group_m") *value = g_asm_group_m;
  else if (n == "last_asm_wgs") *value = g_last_asm_wgs;
  else if (n == "last_asm_rem") *value = g_last_asm_rem;
  else if (n == "last_asm_slices") *value = g_last_asm_slices;
  else if (n == "last_asm_group_m") *value = g_last_asm_group_m;
  else if (n == "asm_fixup_timeouts") *value = asm_fixup_timeouts();
  else if (n == "shard_rccl_ranks") *value = api_shard_rccl_ranks();
  else if (n == "slice_parallel") *value = g_ctx.slice_parallel;
  else if (n == "slice_parallel_min") *value = g_ctx.slice_parallel_min;
  else if (n == "slice_parallel_tiles") *value = g_ctx.slice_parallel_tiles;
  // diagnostics of the last launch (read-only)
  else if (n == "last_f32_config") *value = g_last_f32_cfg;
  else if (n == "last_f32_asm") *value = g_last_f32_asm;
  else if (n == "last_split") *value = g_last_split;
  else if (n == "last_conv_tail") *value = g_last_conv_tail;
  else return fail(LASER_HIP_E_INVALID, "get_option: unknown option '%s'", name);
  return LASER_HIP_OK;
}
int laser_hip_set_shard_devices(int ndev) {  // host-pointer gemm_strided over ndev GPUs (1 = off, 0 = every visible GPU)
  if (ndev < 0 || ndev > kMaxDevices) return fail(LASER_HIP_E_INVALID, "shard devices outside 0..%d", kMaxDevices);
  g_ctx.shard_devices = ndev;
  return LASER_HIP_OK;
}
int laser_hip_get_shard_devices(void) { return g_ctx.shard_devices; }
const char *laser_hip_f32_config_name(int cfg) { return gemm_f32_config_name(cfg); }

#define LH_DEF_GEMM(SFX, T)                                                                                   \
  int laser_hip_gemm_strided_##SFX(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA,         \
                                   int64_t csA, const T *B, int64_t rsB, int64_t csB, T beta, T *C,           \
                                   int64_t rsC, int64_t csC) {                                                \
    return gemm_host<T>(M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC);                         \
  }                                                                                                           \
  int laser_hip_gemm_strided_##SFX##_dev(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA,   \
                                         int64_t csA, const T *B, int64_t rsB, int64_t csB, T beta, T *C,     \
                                         int64_t rsC, int64_t csC, void *stream) {                            \
    return gemm_dev<T>(1, M, N, K, alpha, A, rsA, csA, 0, B, rsB, csB, 0, beta, C, rsC, csC, 0, stream);      \
  }                                                                                                           \
  int laser_hip_gemm_strided_batched_##SFX##_dev(                                                             \
      int64_t batch, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA,          \
      int64_t bsA, const T *B, int64_t rsB, int64_t csB, int64_t bsB, T beta, T *C, int64_t rsC, int64_t csC, \
      int64_t bsC, void *stream) {                                                                            \
    return gemm_dev<T>(batch, M, N, K, alpha, A, rsA, csA, bsA, B, rsB, csB, bsB, beta, C, rsC, csC, bsC,     \
                       stream);                                                                               \
  }                                                                                                           \
  int64_t laser_hip_gemm_prepackA_mem_required_##SFX(int64_t M, int64_t N, int64_t K) {                       \
    return prepack_bytes<T>(true, M, N, K);                                                                   \
  }                                                                                                           \
  int64_t laser_hip_gemm_prepackB_mem_required_##SFX(int64_t M, int64_t N, int64_t K) {                       \
    return prepack_bytes<T>(false, M, N, K);                                                                  \
  }                                                                                                           \
  int laser_hip_gemm_prepackA_##SFX(void *dst, int64_t M, int64_t N, int64_t K, const T *A, int64_t rs,       \
                                    int64_t cs) {                                                             \
    return prepack_host<T>(true, dst, M, N, K, A, rs, cs);                                                    \
  }                                                                                                           \
  int laser_hip_gemm_prepackB_##SFX(void *dst, int64_t M, int64_t N, int64_t K, const T *B, int64_t rs,       \
                                    int64_t cs) {                                                             \
    return prepack_host<T>(false, dst, M, N, K, B, rs, cs);                                                   \
  }                                                                                                           \
  int laser_hip_gemm_packed_##SFX(int64_t M, int64_t N, int64_t K, T alpha, const void *pA, const void *pB,   \
                                  T beta, T *C, int64_t rsC, int64_t csC) {                                   \
    return packed_host<T>(M, N, K, alpha, pA, pB, beta, C, rsC, csC);                                         \
  }                                                                                                           \
  int laser_hip_gemm_prepackA_##SFX##_dev(void *d, int64_t M, int64_t N, int64_t K, const T *A, int64_t rs,   \
                                          int64_t cs, void *stream) {                                         \
    return prepack_dev<T>(true, d, M, N, K, A, rs, cs, stream);                                               \
  }                                                                                                           \
  int laser_hip_gemm_prepackB_##SFX##_dev(void *d, int64_t M, int64_t N, int64_t K, const T *B, int64_t rs,   \
                                          int64_t cs, void *stream) {                                         \
    return prepack_dev<T>(false, d, M, N, K, B, rs, cs, stream);                                              \
  }                                                                                                           \
  int laser_hip_gemm_packed_##SFX##_dev(int64_t M, int64_t N, int64_t K, T alpha, const void *dA,             \
                                        const void *dB, T beta, T *dC, int64_t rsC, int64_t csC,              \
                                        void *stream) {                                                       \
    return packed_dev<T>(M, N, K, alpha, dA, dB, beta, dC, rsC, csC, stream);                                 \
  }
LH_DEF_GEMM(f32, float)
LH_DEF_GEMM(f64, double)
LH_DEF_GEMM(i32, int32_t)
LH_DEF_GEMM(i64, int64_t)
#undef LH_DEF_GEMM

int laser_hip_gemm_prepack_release(void *packed) {
  if (!packed) return fail(LASER_HIP_E_INVALID, "null packed buffer");
  std::lock_guard<std::mutex> lk(g_mu);
  PackHandle h;
  memcpy(&h, packed, sizeof h);
  if (h.magic != kMagic) return fail(LASER_HIP_E_HANDLE, "buffer does not hold a pre-packed operand (never packed, or already released)");
  panel_cache_drop_locked(h.id);      // (copies of the buffer lose their cached device image too: they re-upload on their next use)
  memset(packed, 0, sizeof h);        // the buffer no longer reads as a packed operand
  return LASER_HIP_OK;
}

#define LH_DEF_TR(SFX, ELEM)                                                                                  \
  int laser_hip_transpose2d_copy_##SFX(void *dst, const void *src, int64_t NR, int64_t NC) {                  \
    return transpose_host(dst, src, 1, NR, NC, ELEM);                                                         \
  }                                                                                                           \
  int laser_hip_transpose2d_batched_##SFX(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC) {    \
    return transpose_host(dst, src, N, NR, NC, ELEM);                                                         \
  }                                                                                                           \
  int laser_hip_nchw2nhwc_##SFX(void *dst, const void *src, int64_t N, int64_t C, int64_t H, int64_t W) {     \
    return transpose_host(dst, src, N, C, H * W, ELEM); /* swapaxes.nim:98 */                                 \
  }                                                                                                           \
  int laser_hip_nhwc2nchw_##SFX(void *dst, const void *src, int64_t N, int64_t C, int64_t H, int64_t W) {     \
    return transpose_host(dst, src, N, H * W, C, ELEM); /* swapaxes.nim:112 */                                \
  }                                                                                                           \
  int laser_hip_transpose2d_batched_##SFX##_dev(void *dst, const void *src, int64_t N, int64_t NR,            \
                                                int64_t NC, void *stream) {                                   \
    if (N < 0 || NR < 0 || NC < 0) return fail(LASER_HIP_E_INVALID, "negative dimension");                    \
    if (int rc = ensure_init()) return rc;                                                                    \
    if (N == 0 || NR == 0 || NC == 0) return LASER_HIP_OK;                                                    \
    if (!dst || !src) return fail(LASER_HIP_E_INVALID, "null pointer");                                       \
    HIP_TRY(launch_transpose_batched(dst, src, N, NR, NC, ELEM, (hipStream_t)stream));                        \
    return LASER_HIP_OK;                                                                                      \
  }
LH_DEF_TR(b32, 4)
LH_DEF_TR(b64, 8)
LH_DEF_TR(b16, 2)
LH_DEF_TR(b8, 1)
#undef LH_DEF_TR

int laser_hip_conv2d_out_shape(int64_t iN, int64_t iC, int64_t iH, int64_t iW, int64_t c_out, int64_t c_in,
                               int64_t kH, int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW,
                               int64_t *oN, int64_t *oC, int64_t *oH, int64_t *oW) {
  if (int rc = conv_check(iN, iC, iH, iW, c_out, c_in, kH, kW, pH, pW, sH, sW)) return rc;
  *oN = iN;
  *oC = c_out;
  out_hw(iH, iW, kH, kW, pH, pW, sH, sW, oH, oW);
  return LASER_HIP_OK;
}

int64_t laser_hip_im2col_workspace_size(int64_t iN, int64_t iC, int64_t iH, int64_t iW, int64_t c_out,
                                        int64_t c_in, int64_t kH, int64_t kW, int64_t pH, int64_t pW,
                                        int64_t sH, int64_t sW) {
  (void)iN; (void)c_out; (void)c_in;
  if (sH <= 0 || sW <= 0) return 0;
  int64_t oH, oW;
  out_hw(iH, iW, kH, kW, pH, pW, sH, sW, &oH, &oW);
  return iC * kH * kW * oH * oW;  // conv2d_im2col.nim:19-20
}

// im2col*[T] (conv2d_im2col.nim:42-88) is generic in the element type: pure data movement, one kernel per element size
static int im2col_api_dev(void *dws, int64_t oH, int64_t oW, const void *din, int64_t batch, int64_t iC, int64_t iH, int64_t iW,
                          int64_t kH, int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, int elem, void *stream) {
  if (int rc = ensure_init()) return rc;
  if (!dws || !din) return fail(LASER_HIP_E_INVALID, "null pointer");
  if (oH < 0 || oW < 0 || batch < 0 || iC <= 0 || iH <= 0 || iW <= 0 || kH <= 0 || kW <= 0 || pH < 0 || pW < 0 || sH <= 0 || sW <= 0)
    return fail(LASER_HIP_E_INVALID, "bad shape");
  HIP_TRY(launch_im2col(dws, oH, oW, din, batch, iC, iH, iW, kH, kW, pH, pW, sH, sW, elem, (hipStream_t)stream));
  return LASER_HIP_OK;
}
static int im2col_api_host(void *ws, int64_t oH, int64_t oW, const void *in, int64_t iC, int64_t iH, int64_t iW, int64_t kH, int64_t kW,
                           int64_t pH, int64_t pW, int64_t sH, int64_t sW, int elem) {
  if (int rc = ensure_init()) return rc;
  if (!ws || !in) return fail(LASER_HIP_E_INVALID, "null pointer");
  if (oH < 0 || oW < 0 || iC <= 0 || iH <= 0 || iW <= 0 || kH <= 0 || kW <= 0 || pH < 0 || pW < 0 || sH <= 0 || sW <= 0)
    return fail(LASER_HIP_E_INVALID, "bad shape");
  HostCall hc;
  if (hc.rc) return hc.rc;
  const size_t ib = (size_t)iC * iH * iW * elem, wb = (size_t)iC * kH * kW * oH * oW * elem;
  if (wb == 0) return LASER_HIP_OK;
  void *di, *dw;
  if (int rc = scratch_get(0, ib, &di)) return rc;
  if (int rc = scratch_get(4, wb, &dw)) return rc;
  HIP_TRY(hipMemcpy(di, in, ib, hipMemcpyHostToDevice));
  HIP_TRY(launch_im2col(dw, oH, oW, di, 1, iC, iH, iW, kH, kW, pH, pW, sH, sW, elem, nullptr));
  HIP_TRY(hipMemcpy(ws, dw, wb, hipMemcpyDeviceToHost));
  return LASER_HIP_OK;
}
#define LH_DEF_IM2COL(SFX, T)                                                                                               \
  int laser_hip_im2col_##SFX##_dev(T *dws, int64_t oH, int64_t oW, const T *din, int64_t batch, int64_t iC, int64_t iH,      \
                                   int64_t iW, int64_t kH, int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW,       \
                                   void *stream) {                                                                           \
    return im2col_api_dev(dws, oH, oW, din, batch, iC, iH, iW, kH, kW, pH, pW, sH, sW, (int)sizeof(T), stream);              \
  }                                                                                                                          \
  int laser_hip_im2col_##SFX(T *ws, int64_t oH, int64_t oW, const T *in, int64_t iC, int64_t iH, int64_t iW, int64_t kH,     \
                             int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW) {                                   \
    return im2col_api_host(ws, oH, oW, in, iC, iH, iW, kH, kW, pH, pW, sH, sW, (int)sizeof(T));                              \
  }
LH_DEF_IM2COL(f32, float)
LH_DEF_IM2COL(f64, double)
#undef LH_DEF_IM2COL

static int conv2d_api_dev(float *dout, const float *din, int64_t iN, int64_t iC, int64_t iH,
                                    int64_t iW, const float *dker, int64_t c_out, int64_t c_in, int64_t kH,
                                    int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *dws,
                                    int64_t dws_elems, const float *dbias, int act, void *stream) {
  if (int rc = conv_check(iN, iC, iH, iW, c_out, c_in, kH, kW, pH, pW, sH, sW)) return rc;
  if (int rc = ensure_init()) return rc;
  if (iN == 0) return LASER_HIP_OK;
  if (!dout || !din || !dker) return fail(LASER_HIP_E_INVALID, "null pointer");
  if (iN > 65535) return fail(LASER_HIP_E_INVALID, "batch > 65535");
  hipStream_t s = (hipStream_t)stream;
  const bool direct = (kH * kW == 1) && pH == 0 && pW == 0 && sH == 1 && sW == 1;
  if (direct || conv_takes_implicit(iC, iH, iW, kH, kW, pH, pW, sH, sW))  // nothing is materialised
    return conv_dev(dout, din, iN, iC, iH, iW, dker, c_out, kH, kW, pH, pW, sH, sW, nullptr, s, dbias, act);
  // Explicit im2col path (kernels larger than 8x8, huge images, laser_hip_set_conv_implicit(0)).  The caller's
  // workspace follows the REFERENCE's contract -- one image, im2col_workspace_size elements
  // (conv2d_im2col.nim:19-20, reused between batches :99) -- so it is used as what it is: a caller that hands
  // over iN images' worth gets them all expanded in one pass, one image's worth is looped over image by image,
  // anything smaller is rejected, and no workspace means stream-ordered library scratch (never a buffer shared
  // between streams).
  int64_t oH, oW;
  out_hw(iH, iW, kH, kW, pH, pW, sH, sW, &oH, &oW);
  const int64_t w1 = iC * kH * kW * oH * oW;
  if (dws && dws_elems >= 0 && dws_elems < w1)
    return fail(LASER_HIP_E_INVALID, "im2col workspace holds %lld elements, one image needs %lld", (long long)dws_elems, (long long)w1);
  float *own = nullptr;
  int64_t imgs = iN;  // images expanded per pass
  if (!dws) {
    while (imgs > 1 && (double)imgs * (double)w1 * 4.0 > 2.0e9) imgs = (imgs + 1) / 2;  // bound the scratch
    HIP_TRY(scratch_alloc_async((void **)&own, (size_t)imgs * (size_t)w1 * 4, s));
    dws = own;
  } else {
    // capacity unknown (the plain entry points): the reference's contract is ONE image's worth -> image by image
    imgs = dws_elems >= 0 ? std::max<int64_t>(1, std::min<int64_t>(iN, dws_elems / w1)) : 1;
  }
  int rc = LASER_HIP_OK;
  for (int64_t n0 = 0; n0 < iN && rc == LASER_HIP_OK; n0 += imgs) {
    const int64_t nb = std::min<int64_t>(imgs, iN - n0);
    rc = conv_dev(dout + n0 * c_out * oH * oW, din + n0 * iC * iH * iW, nb, iC, iH, iW, dker, c_out, kH, kW, pH, pW, sH, sW,
                  dws, s, dbias, act);
  }
  if (own) {
    const hipError_t e = hipFreeAsync(own, s);
    if (rc == LASER_HIP_OK && e != hipSuccess) rc = fail(LASER_HIP_E_HIP, "hipFreeAsync: %s", hipGetErrorString(e));
  }
  return rc;
}
int laser_hip_conv2d_im2col_f32_dev(float *dout, const float *din, int64_t iN, int64_t iC, int64_t iH,
                                    int64_t iW, const float *dker, int64_t c_out, int64_t c_in, int64_t kH,
                                    int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *dws,
                                    void *stream) {
  return conv2d_api_dev(dout, din, iN, iC, iH, iW, dker, c_out, c_in, kH, kW, pH, pW, sH, sW, dws, -1, nullptr, 0, stream);
}
int laser_hip_conv2d_im2col_ex_f32_dev(float *dout, const float *din, int64_t iN, int64_t iC, int64_t iH,
                                       int64_t iW, const float *dker, int64_t c_out, int64_t c_in, int64_t kH,
                                       int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *dws,
                                       const float *dbias, int act, void *stream) {
  return conv2d_api_dev(dout, din, iN, iC, iH, iW, dker, c_out, c_in, kH, kW, pH, pW, sH, sW, dws, -1, dbias, act, stream);
}

static int conv2d_api_host(float *out, const float *in, int64_t iN, int64_t iC, int64_t iH, int64_t iW,
                                const float *ker, int64_t c_out, int64_t c_in, int64_t kH, int64_t kW,
                                int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *pworkspace,
                                const float *bias, int act) {
  if (int rc = conv_check(iN, iC, iH, iW, c_out, c_in, kH, kW, pH, pW, sH, sW)) return rc;
  if (int rc = ensure_init()) return rc;
  if (iN == 0) return LASER_HIP_OK;
  if (!out || !in || !ker) return fail(LASER_HIP_E_INVALID, "null pointer");
  if (iN > 65535) return fail(LASER_HIP_E_INVALID, "batch > 65535");
  HostCall hc;
  if (hc.rc) return hc.rc;
  int64_t oH, oW;
  out_hw(iH, iW, kH, kW, pH, pW, sH, sW, &oH, &oW);
  const size_t ib = (size_t)iN * iC * iH * iW * 4, kb = (size_t)c_out * iC * kH * kW * 4;
  const size_t ob = (size_t)iN * c_out * oH * oW * 4, w1 = (size_t)iC * kH * kW * oH * oW * 4;
  void *di, *dk, *dout, *dws;
  const bool direct = (kH * kW == 1) && pH == 0 && pW == 0 && sH == 1 && sW == 1;
  const bool implicit = conv_takes_implicit(iC, iH, iW, kH, kW, pH, pW, sH, sW);
  if (int rc = scratch_get(0, ib, &di)) return rc;
  if (int rc = scratch_get(1, kb, &dk)) return rc;
  if (int rc = scratch_get(2, ob, &dout)) return rc;
  if (int rc = scratch_get(4, implicit ? w1 : w1 * iN, &dws)) return rc;
  HIP_TRY(hipMemcpy(di, in, ib, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dk, ker, kb, hipMemcpyHostToDevice));
  void *dbias = nullptr;
  if (bias) {
    if (int rc = scratch_get(5, (size_t)c_out * 4, &dbias)) return rc;
    HIP_TRY(hipMemcpy(dbias, bias, (size_t)c_out * 4, hipMemcpyHostToDevice));
  }
  if (int rc = conv_dev((float *)dout, (const float *)di, iN, iC, iH, iW, (const float *)dk, c_out, kH, kW, pH,
                        pW, sH, sW, (float *)dws, nullptr, (const float *)dbias, act))
    return rc;
  HIP_TRY(hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost));
  // like the reference, the caller's workspace ends up holding the LAST image's im2col matrix
  if (pworkspace && !direct && w1 > 0) {
    const char *src = (const char *)dws + (size_t)(iN - 1) * w1;
    if (implicit) {  // nothing was materialised: expand just the last image for the caller
      HIP_TRY(launch_im2col_f32((float *)dws, oH, oW, (const float *)di + (size_t)(iN - 1) * iC * iH * iW, 1, iC, iH, iW,
                                kH, kW, pH, pW, sH, sW, nullptr));
      src = (const char *)dws;
    }
    HIP_TRY(hipMemcpy(pworkspace, src, w1, hipMemcpyDeviceToHost));
  }
  return LASER_HIP_OK;
}

int laser_hip_conv2d_im2col_f32(float *out, const float *in, int64_t iN, int64_t iC, int64_t iH, int64_t iW,
                                const float *ker, int64_t c_out, int64_t c_in, int64_t kH, int64_t kW,
                                int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *pworkspace) {
  return conv2d_api_host(out, in, iN, iC, iH, iW, ker, c_out, c_in, kH, kW, pH, pW, sH, sW, pworkspace, nullptr, 0);
}
int laser_hip_conv2d_im2col_ex_f32(float *out, const float *in, int64_t iN, int64_t iC, int64_t iH, int64_t iW,
                                   const float *ker, int64_t c_out, int64_t c_in, int64_t kH, int64_t kW,
                                   int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *pworkspace,
                                   const float *bias, int act) {
  return conv2d_api_host(out, in, iN, iC, iH, iW, ker, c_out, c_in, kH, kW, pH, pW, sH, sW, pworkspace, bias, act);
}

#define LH_DEF_GEMM_EX(SFX, T)                                                                                \
  int laser_hip_gemm_strided_ex_##SFX(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA,      \
                                      int64_t csA, const T *B, int64_t rsB, int64_t csB, T beta, T *C,        \
                                      int64_t rsC, int64_t csC, const T *bias, int64_t rsBias,                \
                                      int64_t csBias, int act) {                                              \
    const Epi<T> e = make_epi<T>(bias, rsBias, csBias, act);                                                  \
    return gemm_host<T>(M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC, &e);                     \
  }                                                                                                           \
  int laser_hip_gemm_strided_ex_##SFX##_dev(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, \
                                            int64_t csA, const T *B, int64_t rsB, int64_t csB, T beta, T *C,  \
                                            int64_t rsC, int64_t csC, const T *bias, int64_t rsBias,          \
                                            int64_t csBias, int act, void *stream) {                          \
    const Epi<T> e = make_epi<T>(bias, rsBias, csBias, act);                                                  \
    return gemm_dev<T>(1, M, N, K, alpha, A, rsA, csA, 0, B, rsB, csB, 0, beta, C, rsC, csC, 0, stream, &e);  \
  }
LH_DEF_GEMM_EX(f32, float)
LH_DEF_GEMM_EX(f64, double)
#undef LH_DEF_GEMM_EX

// ---- device tensor storage -- laser/tensor/allocator.nim, initialization.nim -----------------------
// Freed storages are kept on a per-size free list (exact match of the 256-byte-rounded size, at most
// kStorageCacheMax bytes in total) so that chains of tensor-producing calls do not pay hipMalloc / hipFree -- a
// device allocation costs ~100 us, more than a 2048^3 product.  Reused blocks are zero-filled again: the contract
// stays allocShared0's.  laser_hip_storage_trim() / laser_hip_finalize() release the list.
static int storage_alloc(void **d, int64_t bytes, hipStream_t stream, bool ordered) {
  if (!d || bytes < 0) return fail(LASER_HIP_E_INVALID, "storage_alloc: bad argument");
  if (int rc = ensure_init()) return rc;
  *d = nullptr;
  if (bytes == 0) return LASER_HIP_OK;
  const size_t want = ((size_t)bytes + 255) & ~(size_t)255;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const size_t key = storage_key(want, dev);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_free_storage.find(key);
    if (it != g_free_storage.end() && !it->second.empty()) {
      *d = it->second.back();
      it->second.pop_back();
      g_free_storage_bytes -= want;
    }
  }
  // a recycled block may still be read by work queued on any stream when its owner dropped it (hipFree would have
  // waited for the device; the free list does not): wait here, where it is only paid on reuse
  if (*d) HIP_TRY(hipDeviceSynchronize());
  if (!*d) {
    hipError_t e = hipMalloc(d, want);  // hipMalloc is 256-byte aligned >= LASER_MEM_ALIGN
    if (e != hipSuccess) {             // out of memory: give the cached blocks back and retry once
      storage_trim();
      e = hipMalloc(d, want);
    }
    if (e != hipSuccess) return fail(LASER_HIP_E_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_mu);
    g_live_storage[*d] = key;
  }
  // zero fill (allocShared0): ordered on the caller's stream -- the stream the tensor's first kernel will run on --
  // or, for the plain entry point, completed before returning (PyTorch-style side streams are non-blocking: a fill
  // left running on the NULL stream could land after, or beside, the first kernel that writes the block)
  hipError_t e = hipMemsetAsync(*d, 0, (size_t)bytes, stream);
  if (e == hipSuccess && !ordered) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return fail(LASER_HIP_E_HIP, "zero fill: %s", hipGetErrorString(e));
  return LASER_HIP_OK;
}
int laser_hip_storage_alloc(void **d, int64_t bytes) { return storage_alloc(d, bytes, nullptr, false); }
int laser_hip_storage_alloc_stream(void **d, int64_t bytes, void *stream) { return storage_alloc(d, bytes, (hipStream_t)stream, true); }
int laser_hip_storage_free(void *d) {
  if (!d) return LASER_HIP_OK;
  if (int rc = ensure_init()) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live_storage.find(d);
  if (it == g_live_storage.end()) return fail(LASER_HIP_E_INVALID, "storage_free: not a laser_hip storage");
  const size_t key = it->second, sz = storage_size(key);
  if (g_free_storage_bytes + sz <= kStorageCacheMax) {
    g_free_storage[key].push_back(d);
    g_free_storage_bytes += sz;
    return LASER_HIP_OK;
  }
  g_live_storage.erase(it);
  HIP_TRY(hipFree(d));
  return LASER_HIP_OK;
}
int laser_hip_storage_trim(void) {
  if (int rc = ensure_init()) return rc;
  storage_trim();
  return LASER_HIP_OK;
}
// Host <-> device copies of a storage, ORDERED on `stream` (after the kernels already queued there that produce or
// still read the block) and complete when the call returns.  The plain forms use the NULL stream, which does not
// wait for non-blocking streams: a caller that computes on its own stream passes that stream here.
int laser_hip_storage_upload_stream(void *d, const void *h, int64_t bytes, void *stream) {
  if (bytes < 0 || (bytes > 0 && (!d || !h))) return fail(LASER_HIP_E_INVALID, "storage_upload: bad argument");
  if (int rc = ensure_init()) return rc;
  if (bytes) {
    HIP_TRY(hipMemcpyAsync(d, h, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  }
  return LASER_HIP_OK;
}
int laser_hip_storage_download_stream(void *h, const void *d, int64_t bytes, void *stream) {
  if (bytes < 0 || (bytes > 0 && (!d || !h))) return fail(LASER_HIP_E_INVALID, "storage_download: bad argument");
  if (int rc = ensure_init()) return rc;
  if (bytes) {
    HIP_TRY(hipMemcpyAsync(h, d, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  }
  return LASER_HIP_OK;
}
int laser_hip_storage_upload(void *d, const void *h, int64_t bytes) { return laser_hip_storage_upload_stream(d, h, bytes, nullptr); }
int laser_hip_storage_download(void *h, const void *d, int64_t bytes) { return laser_hip_storage_download_stream(h, d, bytes, nullptr); }
int laser_hip_storage_set_zero(void *d, int64_t bytes, void *stream) {
  if (bytes < 0 || (bytes > 0 && !d)) return fail(LASER_HIP_E_INVALID, "storage_set_zero: bad argument");
  if (int rc = ensure_init()) return rc;
  if (bytes) HIP_TRY(hipMemsetAsync(d, 0, (size_t)bytes, (hipStream_t)stream));
  return LASER_HIP_OK;
}
int laser_hip_copy_strided_b32_dev(void *dst, const int64_t *ds, const void *src, const int64_t *ss,
                                   const int64_t *shape, int rank, void *stream) {
  return copy_strided_api<uint32_t>(dst, ds, src, ss, shape, rank, stream);
}
int laser_hip_copy_strided_b64_dev(void *dst, const int64_t *ds, const void *src, const int64_t *ss,
                                   const int64_t *shape, int rank, void *stream) {
  return copy_strided_api<uint64_t>(dst, ds, src, ss, shape, rank, stream);
}

}  // extern "C"  (a template follows)

// ---- elementwise map over strided views: the device twin of forEach (map_strided.hip) ---------------------------
namespace {
template <typename T>
int map_api(int op, bool binary, T *dst, const int64_t *ds, const T *a, const int64_t *as, const T *b, const int64_t *bs,
            const int64_t *shape, int rank, T alpha, T beta, void *stream) {
  if (rank < 0 || rank > kMaxRank) return fail(LASER_HIP_E_INVALID, "rank %d outside 0..%d (LASER_MAXRANK)", rank, kMaxRank);
  const bool is_bin = op >= LASER_HIP_MAP_ADD && op <= LASER_HIP_MAP_AXPBY && op != 35;   // (35: the removed DIV -- never a silent copy)
  const bool is_un = op >= LASER_HIP_MAP_COPY && op <= LASER_HIP_MAP_SQUARE;
  if (binary ? !is_bin : !is_un) return fail(LASER_HIP_E_INVALID, "map op %d is not a %s op", op, binary ? "binary" : "unary");
  const int nin = binary ? 2 : (op == LASER_HIP_MAP_FILL ? 0 : 1);
  if (rank > 0 && (!ds || !shape || (nin >= 1 && !as) || (nin >= 2 && !bs))) return fail(LASER_HIP_E_INVALID, "null shape/strides");
  int64_t total = 1;
  for (int d = 0; d < rank; d++) {
    if (shape[d] < 0) return fail(LASER_HIP_E_INVALID, "negative extent");
    total *= shape[d];
  }
  if (int rc = ensure_init()) return rc;
  if (total == 0) return LASER_HIP_OK;
  if (!dst || (nin >= 1 && !a) || (nin >= 2 && !b)) return fail(LASER_HIP_E_INVALID, "null buffer");
  HIP_TRY(launch_map_strided<T>(op, nin, dst, ds, a, as, b, bs, shape, rank, alpha, beta, (hipStream_t)stream));
  return LASER_HIP_OK;
}
}  // namespace

extern "C" {

#define LH_DEF_MAP(SFX, T)                                                                                              \
  int laser_hip_map_strided_unary_##SFX##_dev(int op, T *dst, const int64_t *ds, const T *a, const int64_t *as,          \
                                              const int64_t *shape, int rank, T alpha, T beta, void *stream) {           \
    return map_api<T>(op, false, dst, ds, a, as, nullptr, nullptr, shape, rank, alpha, beta, stream);                    \
  }                                                                                                                     \
  int laser_hip_map_strided_binary_##SFX##_dev(int op, T *dst, const int64_t *ds, const T *a, const int64_t *as,         \
                                               const T *b, const int64_t *bs, const int64_t *shape, int rank,            \
                                               T alpha, T beta, void *stream) {                                         \
    return map_api<T>(op, true, dst, ds, a, as, b, bs, shape, rank, alpha, beta, stream);                                \
  }
LH_DEF_MAP(f32, float)
LH_DEF_MAP(f64, double)
LH_DEF_MAP(i32, int32_t)
LH_DEF_MAP(i64, int64_t)
#undef LH_DEF_MAP

// ---- pinned host memory for the host-pointer entry points -----------------------------------------------------------
// Laser leaves buffer management to the caller ("creating or reusing buffers is left at the discretion of the
// high-level lib", Design.md:5-7).  The host-pointer entry points work on any memory; from PAGEABLE memory every
// hipMemcpy is staged or pinned on the fly by the runtime.  A tensor allocator that wants the full PCIe rate allocates
// its buffers here (or registers what it already has): the same entry points then run their uploads / downloads as
// direct asynchronous DMA.  Nothing else changes; results are identical.
int laser_hip_host_alloc(void **host_ptr, int64_t bytes) {
  if (!host_ptr || bytes < 0) return fail(LASER_HIP_E_INVALID, "host_alloc: bad argument");
  if (int rc = ensure_init()) return rc;
  *host_ptr = nullptr;
  if (bytes == 0) return LASER_HIP_OK;
  HIP_TRY(hipHostMalloc(host_ptr, (size_t)bytes, hipHostMallocDefault));
  return LASER_HIP_OK;
}
int laser_hip_host_free(void *host_ptr) {
  if (!host_ptr) return LASER_HIP_OK;
  HIP_TRY(hipHostFree(host_ptr));
  return LASER_HIP_OK;
}
int laser_hip_host_register(void *host_ptr, int64_t bytes) {
  if (!host_ptr || bytes <= 0) return fail(LASER_HIP_E_INVALID, "host_register: bad argument");
  if (int rc = ensure_init()) return rc;
  HIP_TRY(hipHostRegister(host_ptr, (size_t)bytes, hipHostRegisterDefault));
  return LASER_HIP_OK;
}
int laser_hip_host_unregister(void *host_ptr) {
  if (!host_ptr) return fail(LASER_HIP_E_INVALID, "host_unregister: null pointer");
  HIP_TRY(hipHostUnregister(host_ptr));
  return LASER_HIP_OK;
}

int laser_hip_cblas_sgemm(int order, int tA, int tB, int64_t M, int64_t N, int64_t K, float alpha, const float *A,
                          int64_t lda, const float *B, int64_t ldb, float beta, float *C, int64_t ldc) {
  return cblas_gemm<float>(order, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}
int laser_hip_cblas_dgemm(int order, int tA, int tB, int64_t M, int64_t N, int64_t K, double alpha,
                          const double *A, int64_t lda, const double *B, int64_t ldb, double beta, double *C,
                          int64_t ldc) {
  return cblas_gemm<double>(order, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

}  // extern "C"
