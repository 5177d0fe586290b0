// laser_amd/csrc/sharded.cpp -- row-panel sharded gemm_strided over the GPUs of ONE node, in ONE process, behind the
// C-ABI (include/laser_hip.h, "sharded" section).
//
// Why this shape: Laser itself partitions M across its OpenMP threads with no cross-thread reduction
// (gemm.nim:160-176: `omp for` over the ic row blocks), so output rows are independent units.  Each GPU owns row
// panels of A, all of B (replicated) and produces the matching row panels of C; there is no K split, hence no
// reduction, and the per-element arithmetic -- including Laser's kc-slice accumulation order -- is exactly the
// single-GPU one: results are bit-identical whatever the device count.
//
//   host-pointer form   the drop-in: the caller's gemm_strided(M, N, K, ..., host pointers) is cut into one
//                       contiguous row range per GPU; every GPU runs the ordinary host-pointer pipeline (upload B,
//                       stream A panels, download C panels) on its own PCIe link, one host thread per GPU.  C goes
//                       straight back to host memory: no inter-GPU traffic at all.
//   device-resident     operands already in HBM (what the roofline is measured on): rows are dealt block-cyclically
//                       (panels_per_dev sub-panels per GPU), and every GPU ends up with ALL of C -- the all-gather
//                       of C over xGMI.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so gathering 7/8 of a
//                       2 GiB C is milliseconds, the same order as the 8192^3 product itself; the exchange is
//                       therefore pipelined: as soon as sub-panel s is computed it is sent while sub-panel s+1
//                       multiplies.  Two transports:
//                         PEER  (default) the owner pushes its finished rows to the 7 peers with
//                               hipMemcpyPeerAsync on 7 copy streams -- all links at once, SDMA engines, no
//                               compute units taken from the GEMM;
//                         RCCL  ncclAllGather per slab on a communicator made by ncclCommInitAll (librccl.so is
//                               dlopen'ed on first use, liblaser_hip.so does not link it); its kernels share the
//                               CUs with the GEMM, so the 128x128 assembly tile can be pinned for the local products.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/laser_hip.h"
#include "capi_internal.h"

using namespace laser_hip;

namespace {

constexpr int kMaxRanks = 16;

#define SH_TRY(expr)                                                                                                  \
  do {                                                                                                                \
    hipError_t e_ = (expr);                                                                                           \
    if (e_ != hipSuccess)                                                                                             \
      return api_fail(LASER_HIP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
  } while (0)

struct Plan {
  int64_t M = 0, rows = 0;
  int ndev = 1, ppd = 1;
  int64_t padded_M() const { return rows * ndev * ppd; }
  void panel(int s, int g, int64_t *start, int64_t *valid) const {
    *start = ((int64_t)s * ndev + g) * rows;
    *valid = std::max<int64_t>(0, std::min<int64_t>(rows, M - *start));
  }
};

// Block-cyclic deal of M rows: ndev * ppd panels of `rows` rows, rows a multiple of 256 (the largest workgroup tile)
// when M allows it so every panel runs the vector loaders; no all-empty steps.  (Same rule as
// laser_amd/distributed.py: make_plan -- the per-process RCCL path -- so both paths lay C out identically.)
Plan make_plan(int64_t M, int ndev, int ppd) {
  Plan p;
  p.M = M;
  p.ndev = std::max(1, ndev);
  p.ppd = std::max(1, ppd);
  const int64_t n = (int64_t)p.ndev * p.ppd;
  p.rows = (M + n - 1) / n;
  if (p.rows >= 256) p.rows = (p.rows + 255) / 256 * 256;
  if (p.rows < 1) p.rows = 1;
  while (p.ppd > 1 && p.rows * p.ndev * (p.ppd - 1) >= M) p.ppd--;
  return p;
}

// Per-RANK streams and events (a rank = one entry of the caller's device list; tests may list one physical device
// several times, so this is keyed by rank slot, not by device ordinal).
struct RankCtx {
  int device = -1;
  hipStream_t comp = nullptr, comm = nullptr;
  hipStream_t peer[kMaxRanks] = {};
  std::vector<hipEvent_t> ev;  // one per sub-panel, grown on demand
};
RankCtx g_rank[kMaxRanks];
std::mutex g_shard_mu;  // one sharded call at a time (they use every GPU anyway)

// One persistent host thread per rank slot (created on first use, parked on a condition variable between calls): a sharded call
// hands every slot its job and waits for all of them -- no thread is created or joined per call (that alone cost ~0.1 ms of an
// 8192^3 step in round 3), and a slot's thread keeps its device context, its thread-local scratch and its tile pin warm.
class WorkerPool {
 public:
  template <typename F>
  void run(int n, F &&fn) {
    if (n == 1) {
      fn(0);
      return;
    }
    std::unique_lock<std::mutex> lk(mu_);
    while ((int)threads_.size() < n) {
      const int slot = (int)threads_.size();
      seen_.push_back(gen_);
      threads_.emplace_back([this, slot] { loop(slot); });
    }
    job_ = [&fn](int g) { fn(g); };
    njobs_ = n;
    pending_ = n;
    gen_++;
    cv_.notify_all();
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto &t : threads_)
      if (t.joinable()) t.join();
  }

 private:
  void loop(int slot) {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || (gen_ != seen_[slot] && slot < njobs_); });
      if (stop_) return;
      seen_[slot] = gen_;
      auto job = job_;
      lk.unlock();
      job(slot);
      lk.lock();
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;
  std::vector<uint64_t> seen_;
  std::function<void(int)> job_;
  uint64_t gen_ = 0;
  int njobs_ = 0, pending_ = 0;
  bool stop_ = false;
};
WorkerPool &pool() {
  static WorkerPool *p = new WorkerPool();   // (never destroyed: worker threads must not be joined from a static destructor at exit)
  return *p;
}

// a rank whose call failed may still have work queued that names the caller's buffers: its streams are dropped (the runtime
// releases them when that work has drained) and recreated by the next call
void rank_reset(int r) {
  RankCtx &R = g_rank[r];
  if (R.device < 0) return;
  (void)hipSetDevice(R.device);
  if (R.comp) (void)hipStreamDestroy(R.comp);
  if (R.comm) (void)hipStreamDestroy(R.comm);
  for (auto &p : R.peer)
    if (p) (void)hipStreamDestroy(p), p = nullptr;
  for (auto e : R.ev) (void)hipEventDestroy(e);
  R.ev.clear();
  R.comp = R.comm = nullptr;
  R.device = -1;
}

int rank_setup(int r, int device, int ndev, int nev) {
  RankCtx &R = g_rank[r];
  SH_TRY(hipSetDevice(device));
  if (R.device != device) {  // (re)create on the right device
    if (R.comp) {
      (void)hipStreamDestroy(R.comp);
      (void)hipStreamDestroy(R.comm);
      for (auto &p : R.peer)
        if (p) (void)hipStreamDestroy(p), p = nullptr;
      for (auto e : R.ev) (void)hipEventDestroy(e);
      R.ev.clear();
      R.comp = R.comm = nullptr;
    }
    SH_TRY(hipStreamCreateWithFlags(&R.comp, hipStreamNonBlocking));
    SH_TRY(hipStreamCreateWithFlags(&R.comm, hipStreamNonBlocking));
    R.device = device;
  }
  for (int p = 0; p < ndev; p++)
    if (!R.peer[p]) SH_TRY(hipStreamCreateWithFlags(&R.peer[p], hipStreamNonBlocking));
  while ((int)R.ev.size() < nev) {
    hipEvent_t e;
    SH_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    R.ev.push_back(e);
  }
  return LASER_HIP_OK;
}

int device_list(int ndev, const int *devices, std::vector<int> *out) {
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return api_fail(LASER_HIP_E_NODEVICE, "no HIP device available");
  if (ndev <= 0) ndev = have;  // "all of them"
  if (ndev > kMaxRanks) return api_fail(LASER_HIP_E_INVALID, "more than %d devices", kMaxRanks);
  out->resize(ndev);
  for (int g = 0; g < ndev; g++) {
    const int d = devices ? devices[g] : g;
    if (d < 0 || d >= have) return api_fail(LASER_HIP_E_INVALID, "device %d out of range (%d devices)", d, have);
    (*out)[g] = d;
  }
  return LASER_HIP_OK;
}

void enable_peers(const std::vector<int> &dev) {
  static std::mutex mu;
  static bool done[kMaxRanks][kMaxRanks] = {};
  std::lock_guard<std::mutex> lk(mu);
  for (int a : dev)
    for (int b : dev) {
      if (a == b || a >= kMaxRanks || b >= kMaxRanks || done[a][b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can && hipSetDevice(a) == hipSuccess) {
        const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
        if (e != hipSuccess) (void)hipGetLastError();  // already enabled: fine
      }
      done[a][b] = true;  // (without peer access hipMemcpyPeerAsync still works, staged through the host)
    }
}

// ---- RCCL, loaded on demand ------------------------------------------------------------------------------------
typedef void *ncclComm_t_;
struct Rccl {
  void *h = nullptr;
  int (*CommInitAll)(ncclComm_t_ *, int, const int *) = nullptr;
  int (*CommDestroy)(ncclComm_t_) = nullptr;
  int (*CommAbort)(ncclComm_t_) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*CommCount)(ncclComm_t_, int *) = nullptr;      // optional: what the communicator itself says its size is
  std::vector<int> devs;  // the device list the cached communicators were made for
  std::vector<ncclComm_t_> comms;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.h) return LASER_HIP_OK;
  void *h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return api_fail(LASER_HIP_E_INVALID, "gather mode RCCL: cannot load librccl.so (%s)", dlerror());
#define LD(field, name)                                                                          \
  *(void **)(&g_rccl.field) = dlsym(h, name);                                                    \
  if (!g_rccl.field) return api_fail(LASER_HIP_E_INVALID, "librccl.so has no symbol %s", name);
  LD(CommInitAll, "ncclCommInitAll")
  LD(CommDestroy, "ncclCommDestroy")
  LD(CommAbort, "ncclCommAbort")
  LD(AllGather, "ncclAllGather")
  LD(GroupStart, "ncclGroupStart")
  LD(GroupEnd, "ncclGroupEnd")
  LD(GetErrorString, "ncclGetErrorString")
#undef LD
  *(void **)(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");
  g_rccl.h = h;
  return LASER_HIP_OK;
}

int rccl_comms(const std::vector<int> &dev) {
  if (int rc = rccl_load()) return rc;
  if (g_rccl.devs == dev && !g_rccl.comms.empty()) return LASER_HIP_OK;
  for (auto c : g_rccl.comms) (void)g_rccl.CommDestroy(c);
  g_rccl.comms.assign(dev.size(), nullptr);
  g_rccl.devs.clear();
  const int r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)dev.size(), dev.data());
  if (r != 0) {
    g_rccl.comms.clear();
    return api_fail(LASER_HIP_E_HIP, "ncclCommInitAll over %d devices failed: %s", (int)dev.size(), g_rccl.GetErrorString(r));
  }
  g_rccl.devs = dev;
  return LASER_HIP_OK;
}

}  // namespace
// option "shard_rccl_ranks" (read-only diagnostics): the number of ranks of the RCCL communicator the last GATHER_RCCL call used, as
// the communicator reports it (ncclCommCount; the number of communicators made when the library lacks that symbol); 0 = none yet.
// bench.py's N > 1 lines quote it, so that the line itself says whether the all-gather north_star names really spanned N GPUs.
int64_t laser_hip::api_shard_rccl_ranks() {
  if (g_rccl.comms.empty() || !g_rccl.comms[0]) return 0;
  int n = 0;
  if (g_rccl.CommCount && g_rccl.CommCount(g_rccl.comms[0], &n) == 0) return n;
  return (int64_t)g_rccl.comms.size();
}
namespace {

// A rank failed (or the bounded wait expired) while collectives may still be pending: abort every communicator so no
// peer stays blocked inside ncclAllGather, and drop the cache -- the next call builds fresh communicators.
void rccl_abort_all() {
  for (auto c : g_rccl.comms)
    if (c) (void)g_rccl.CommAbort(c);
  g_rccl.comms.clear();
  g_rccl.devs.clear();
}

// Bounded wait for a stream that may carry collectives: hipStreamSynchronize can block forever when a peer never enters one.
// LASER_HIP_SHARD_TIMEOUT_S (default 120) seconds, then hipErrorTimeout.  Streams that carry only this rank's own kernels and
// copies (no transport, peer pushes) cannot hang on a peer: those are waited for with hipStreamSynchronize, which returns the
// moment the work is done (the 50-us poll below cost up to 0.7 % of an 8192^3 step).
hipError_t sync_bounded(hipStream_t s, bool collectives = true) {
  if (!collectives) return hipStreamSynchronize(s);
  static const double limit = [] {
    const char *e = getenv("LASER_HIP_SHARD_TIMEOUT_S");
    const double v = e ? atof(e) : 120.0;
    return v > 0 ? v : 120.0;
  }();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return hipErrorLaunchTimeOut;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclInt32 = 2, kNcclInt64 = 4;  // ncclDataType_t (rccl.h)
template <typename T> struct NcclType;
template <> struct NcclType<float> { static constexpr int v = kNcclFloat32; };
template <> struct NcclType<double> { static constexpr int v = kNcclFloat64; };
template <> struct NcclType<int32_t> { static constexpr int v = kNcclInt32; };
template <> struct NcclType<int64_t> { static constexpr int v = kNcclInt64; };

// ---- typed forwarders to the single-GPU entry points ---------------------------------------------------------------
template <typename T> struct Api;
#define LH_API(T, SFX)                                                                                                \
  template <> struct Api<T> {                                                                                        \
    static int dev(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *B,        \
                   int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC, void *s) {                       \
      return laser_hip_gemm_strided_##SFX##_dev(M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC, s);      \
    }                                                                                                                 \
    static int host(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *B,       \
                    int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC) {                               \
      return laser_hip_gemm_strided_##SFX(M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC);               \
    }                                                                                                                 \
  };
LH_API(float, f32)
LH_API(double, f64)
LH_API(int32_t, i32)
LH_API(int64_t, i64)
#undef LH_API

// ---- device-resident: block-cyclic panels + gather of C ---------------------------------------------------------
template <typename T>
int sharded_dev(int ndev_in, const int *devices, int64_t M, int64_t N, int64_t K, T alpha, const T *const *dA,
                int64_t rsA, int64_t csA, const T *const *dB, int64_t rsB, int64_t csB, T beta, T *const *dC,
                int64_t rsC, int panels_per_dev, int gather, int flags) {
  if (M < 0 || N < 0 || K < 0) return api_fail(LASER_HIP_E_INVALID, "negative dimension");
  if (gather < LASER_HIP_GATHER_NONE || gather > LASER_HIP_GATHER_RCCL) return api_fail(LASER_HIP_E_INVALID, "unknown gather mode %d", gather);
  if (!dA || !dB || !dC) return api_fail(LASER_HIP_E_INVALID, "null pointer table");
  if (int rc = api_ensure_init()) return rc;
  std::vector<int> dev;
  if (int rc = device_list(ndev_in, devices, &dev)) return rc;
  const int ndev = (int)dev.size();
  if (M == 0 || N == 0 || K == 0) return LASER_HIP_OK;  // K == 0: C untouched (gemm.nim:150)
  if (rsC < N) return api_fail(LASER_HIP_E_INVALID, "sharded C is row-major: rowStrideC (%lld) < N (%lld)", (long long)rsC, (long long)N);
  if (gather == LASER_HIP_GATHER_RCCL && rsC != N)
    return api_fail(LASER_HIP_E_INVALID, "gather mode RCCL needs a dense C (rowStrideC == N): slabs are sent as flat buffers");
  for (int g = 0; g < ndev; g++)
    if (!dA[g] || !dB[g] || !dC[g]) return api_fail(LASER_HIP_E_INVALID, "null operand pointer for device slot %d", g);
  const Plan plan = make_plan(M, ndev, panels_per_dev);
  std::lock_guard<std::mutex> lk(g_shard_mu);
  int prev_dev = 0;
  (void)hipGetDevice(&prev_dev);
  if (gather == LASER_HIP_GATHER_PEER && ndev > 1) enable_peers(dev);
  // (RCCL also with ONE rank: a 1-rank communicator runs the same dlopen / ncclCommInitAll / in-place pointer arithmetic
  // as 8 ranks do, so a single-GPU box exercises this transport)
  if (gather == LASER_HIP_GATHER_RCCL) {
    if (int rc = rccl_comms(dev)) return rc;
  }
  for (int g = 0; g < ndev; g++)
    if (int rc = rank_setup(g, dev[g], ndev, plan.ppd)) {
      (void)hipSetDevice(prev_dev);
      return rc;
    }
  // the local products may be pinned to the 128x128 tile (RCCL's kernels hold CUs while the next panel multiplies)
  // The pin is per CALL: every worker thread sets a thread-local override that only its own launches read -- the
  // process-wide configuration (and any concurrent caller's kernel choice) is never touched.
  // (round 5: the pin selects the hand-scheduled `lh_f32_*_128x128x16` ASSEMBLY kernels -- tile class 2 of option "asm_tile" -- not
  // the compiler-scheduled 128x128 configuration it used to force, which gave away 10-15 % per GPU before a byte crossed xGMI)
  const bool pin = (flags & LASER_HIP_SHARD_PIN_TILE) != 0 && std::is_same<T, float>::value;
  const int pin_tile = 2;
  std::atomic<bool> failed{false};

  std::vector<int> rc(ndev, LASER_HIP_OK);
  std::vector<std::string> msg(ndev);
  auto worker = [&](int g) {
    // A failing rank records its error and KEEPS GOING through the remaining collectives (without its products): its
    // peers are inside the same sequence of ncclAllGather calls and would otherwise wait for it forever.  The call as a
    // whole reports the error; C is then unspecified.
    auto bail = [&](int code) {
      if (rc[g] == LASER_HIP_OK) {
        rc[g] = code;
        msg[g] = laser_hip_last_error();
      }
      failed = true;
    };
    RankCtx &R = g_rank[g];
    hipError_t e = hipSetDevice(dev[g]);
    const bool no_device = e != hipSuccess;      // then no stream / event call below may run (it would land on whatever device is current)
    if (no_device) bail(api_fail(LASER_HIP_E_HIP, "hipSetDevice(%d): %s", dev[g], hipGetErrorString(e)));
    if (pin) api_set_thread_asm_tile(pin_tile);
    // one device and nothing to exchange: the panels are one contiguous matrix -- one product, no per-panel launch boundary
    const bool whole = ndev == 1 && gather != LASER_HIP_GATHER_RCCL && !no_device;
    if (whole) {
      const int r = Api<T>::dev(M, N, K, alpha, dA[0], rsA, csA, dB[0], rsB, csB, beta, dC[0], rsC, 1, R.comp);
      if (r != LASER_HIP_OK) bail(r);
    }
    for (int s = 0; s < plan.ppd && !whole; s++) {
      int64_t start, valid;
      plan.panel(s, g, &start, &valid);
      if (no_device) {      // keep the collective sequence alive for the peers (RCCL), touch nothing else
        if (gather != LASER_HIP_GATHER_RCCL) break;
        T *slab = dC[g] + (int64_t)s * ndev * plan.rows * N;
        if (g_rccl.AllGather(slab + (int64_t)g * plan.rows * N, slab, (size_t)plan.rows * N, NcclType<T>::v, g_rccl.comms[g], R.comm) != 0) break;
        continue;
      }
      if (valid > 0 && rc[g] == LASER_HIP_OK) {
        const int r = Api<T>::dev(valid, N, K, alpha, dA[g] + (int64_t)s * plan.rows * rsA, rsA, csA, dB[g], rsB, csB, beta,
                                  dC[g] + start * rsC, rsC, 1, R.comp);
        if (r != LASER_HIP_OK) bail(r);
      }
      if (gather == LASER_HIP_GATHER_NONE || (ndev == 1 && gather != LASER_HIP_GATHER_RCCL)) continue;
      e = hipEventRecord(R.ev[s], R.comp);
      if (e != hipSuccess) bail(api_fail(LASER_HIP_E_HIP, "hipEventRecord: %s", hipGetErrorString(e)));
      if (gather == LASER_HIP_GATHER_PEER) {
        if (valid <= 0 || rc[g] != LASER_HIP_OK) continue;
        for (int p = 0; p < ndev; p++) {
          if (p == g) continue;
          e = hipStreamWaitEvent(R.peer[p], R.ev[s], 0);
          // push my finished rows into peer p's C (rows are contiguous runs of N elements, rowStrideC apart)
          if (e == hipSuccess) {
            if (rsC == N)
              e = hipMemcpyPeerAsync(dC[p] + start * rsC, dev[p], dC[g] + start * rsC, dev[g], (size_t)valid * N * sizeof(T), R.peer[p]);
            else
              e = hipMemcpy2DAsync(dC[p] + start * rsC, (size_t)rsC * sizeof(T), dC[g] + start * rsC, (size_t)rsC * sizeof(T),
                                   (size_t)N * sizeof(T), (size_t)valid, hipMemcpyDeviceToDevice, R.peer[p]);
          }
          if (e != hipSuccess) {
            bail(api_fail(LASER_HIP_E_HIP, "peer copy %d -> %d: %s", dev[g], dev[p], hipGetErrorString(e)));
            break;
          }
        }
      } else {  // RCCL: in-place all-gather of slab s (sub-panel s of every rank = one contiguous block of C)
        e = hipStreamWaitEvent(R.comm, R.ev[s], 0);
        if (e != hipSuccess) bail(api_fail(LASER_HIP_E_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)));
        T *slab = dC[g] + (int64_t)s * ndev * plan.rows * N;
        const int r = g_rccl.AllGather(slab + (int64_t)g * plan.rows * N, slab, (size_t)plan.rows * N, NcclType<T>::v,
                                       g_rccl.comms[g], R.comm);
        if (r != 0) {  // the collective itself is broken: nobody may wait for it
          bail(api_fail(LASER_HIP_E_HIP, "ncclAllGather (rank %d): %s", g, g_rccl.GetErrorString(r)));
          break;
        }
      }
    }
    if (pin) api_set_thread_asm_tile(-2);
    if (no_device) return;
    // this rank's GEMMs and everything it sent; bounded where a peer that never arrives could hang the caller (RCCL)
    const bool coll = gather == LASER_HIP_GATHER_RCCL && ndev > 1;
    e = sync_bounded(R.comp, coll);
    if (e == hipSuccess && ndev > 1 && gather == LASER_HIP_GATHER_PEER)
      for (int p = 0; p < ndev && e == hipSuccess; p++)
        if (p != g) e = sync_bounded(R.peer[p], false);
    if (e == hipSuccess && gather == LASER_HIP_GATHER_RCCL) e = sync_bounded(R.comm, coll);
    if (e != hipSuccess) bail(api_fail(LASER_HIP_E_HIP, "synchronising device %d: %s", dev[g], hipGetErrorString(e)));
  };
  pool().run(ndev, worker);
  if (failed) {
    if (gather == LASER_HIP_GATHER_RCCL) rccl_abort_all();  // pending collectives of a failed call must not outlive it
    // (queued work of a failed call may still name the caller's buffers: they must stay alive until the device is idle --
    // include/laser_hip.h says so; the slots' streams are dropped here and recreated by the next call)
    for (int g = 0; g < ndev; g++) rank_reset(g);
  }
  (void)hipSetDevice(prev_dev);
  for (int g = 0; g < ndev; g++)
    if (rc[g] != LASER_HIP_OK) return api_fail(rc[g], "device slot %d: %s", g, msg[g].c_str());
  return LASER_HIP_OK;
}

// ---- host pointers: one contiguous row range per GPU, each through the ordinary host-pointer pipeline ---------------
template <typename T>
int sharded_host(int ndev_in, const int *devices, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA,
                 int64_t csA, const T *B, int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC) {
  if (M < 0 || N < 0 || K < 0) return api_fail(LASER_HIP_E_INVALID, "negative dimension");
  if (int rc = api_ensure_init()) return rc;
  std::vector<int> dev;
  if (int rc = device_list(ndev_in, devices, &dev)) return rc;
  if (M == 0 || N == 0 || K == 0) return LASER_HIP_OK;
  if (!A || !B || !C) return api_fail(LASER_HIP_E_INVALID, "null operand pointer");
  int ndev = (int)dev.size();
  // Every worker stages the memory SPAN of its share of C (upload when the span has gaps it does not own, copy the
  // whole span back): two workers' spans must therefore never overlap.  Row ranges are disjoint in memory only when the
  // rows of C do not interleave (row-major-like: rsC >= |csC| * (N - 1) + 1); a column-major-like C (csC >= |rsC| *
  // (M - 1) + 1) is cut into COLUMN ranges instead -- B and C by columns, A replicated: columns of C are as independent
  // as rows (no K split either way, so the arithmetic does not change); anything else (overlapping or negative strides
  // on both axes) runs on one device.
  const auto iabs = [](int64_t v) { return v < 0 ? -v : v; };
  const bool by_rows = rsC > 0 && rsC >= iabs(csC) * (N - 1) + 1;
  const bool by_cols = !by_rows && csC > 0 && csC >= iabs(rsC) * (M - 1) + 1;
  if (!by_rows && !by_cols) ndev = 1;
  const int64_t extent = by_cols ? N : M;
  // whole 256-wide tiles per GPU; fewer GPUs when the extent is small
  int64_t rows = (extent + ndev - 1) / ndev;
  if (rows >= 256) rows = (rows + 255) / 256 * 256;
  ndev = (int)std::min<int64_t>(ndev, (extent + rows - 1) / rows);
  std::vector<int> rc(ndev, LASER_HIP_OK);
  std::vector<std::string> msg(ndev);
  auto worker = [&](int g) {
    const int64_t r0 = (int64_t)g * rows, r1 = std::min<int64_t>(extent, r0 + rows);
    api_set_thread_device(dev[g]);  // the host-pointer entry point below runs on this GPU, with its own scratch / streams
    if (by_cols)
      rc[g] = Api<T>::host(M, r1 - r0, K, alpha, A, rsA, csA, B + r0 * csB, rsB, csB, beta, C + r0 * csC, rsC, csC);
    else
      rc[g] = Api<T>::host(r1 - r0, N, K, alpha, A + r0 * rsA, rsA, csA, B, rsB, csB, beta, C + r0 * rsC, rsC, csC);
    if (rc[g] != LASER_HIP_OK) msg[g] = laser_hip_last_error();
    api_set_thread_device(-1);
  };
  {
    std::lock_guard<std::mutex> lk(g_shard_mu);      // (the pool runs one call at a time)
    pool().run(ndev, worker);
  }
  for (int g = 0; g < ndev; g++)
    if (rc[g] != LASER_HIP_OK) return api_fail(rc[g], "device %d: %s", dev[g], msg[g].c_str());
  return LASER_HIP_OK;
}

}  // namespace

namespace laser_hip {
// called by the plain host-pointer gemm_strided when laser_hip_set_shard_devices(n != 1) routes large problems here
template <typename T>
int api_sharded_host(int ndev, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *B,
                     int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC) {
  return sharded_host<T>(ndev, nullptr, M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC);
}
#define LH_INST(T)                                                                                                     \
  template int api_sharded_host<T>(int, int64_t, int64_t, int64_t, T, const T *, int64_t, int64_t, const T *, int64_t, \
                                   int64_t, T, T *, int64_t, int64_t);
LH_INST(float)
LH_INST(double)
LH_INST(int32_t)
LH_INST(int64_t)
#undef LH_INST
}  // namespace laser_hip

extern "C" {

int laser_hip_shard_plan(int64_t M, int ndev, int panels_per_dev, int64_t *rows_per_panel, int *panels_per_dev_used,
                         int64_t *padded_M) {
  if (M < 0 || ndev < 1 || ndev > kMaxRanks) return api_fail(LASER_HIP_E_INVALID, "shard_plan: bad argument");
  const Plan p = make_plan(M, ndev, panels_per_dev);
  if (rows_per_panel) *rows_per_panel = p.rows;
  if (panels_per_dev_used) *panels_per_dev_used = p.ppd;
  if (padded_M) *padded_M = p.padded_M();
  return LASER_HIP_OK;
}

#define LH_DEF_SHARDED(SFX, T)                                                                                        \
  int laser_hip_gemm_strided_##SFX##_sharded(int ndev, const int *devices, int64_t M, int64_t N, int64_t K, T alpha,  \
                                             const T *A, int64_t rsA, int64_t csA, const T *B, int64_t rsB,          \
                                             int64_t csB, T beta, T *C, int64_t rsC, int64_t csC) {                  \
    return sharded_host<T>(ndev, devices, M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC);              \
  }                                                                                                                   \
  int laser_hip_gemm_strided_##SFX##_sharded_dev(int ndev, const int *devices, int64_t M, int64_t N, int64_t K,      \
                                                 T alpha, const T *const *dA_panels, int64_t rsA, int64_t csA,       \
                                                 const T *const *dB, int64_t rsB, int64_t csB, T beta,               \
                                                 T *const *dC, int64_t rsC, int panels_per_dev, int gather,          \
                                                 int flags) {                                                        \
    return sharded_dev<T>(ndev, devices, M, N, K, alpha, dA_panels, rsA, csA, dB, rsB, csB, beta, dC, rsC,            \
                          panels_per_dev, gather, flags);                                                             \
  }
LH_DEF_SHARDED(f32, float)
LH_DEF_SHARDED(f64, double)
LH_DEF_SHARDED(i32, int32_t)
LH_DEF_SHARDED(i64, int64_t)
#undef LH_DEF_SHARDED

}  // extern "C"
