// laser_amd/csrc/common.h -- shared host/device declarations of liblaser_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace laser_hip {

// Stream-ordered scratch (packing passes, limb planes, slice partials): hipMallocFromPoolAsync on a pool THIS LIBRARY owns (one per
// device, made on first use, destroyed by laser_hip_finalize), with a release threshold of 4 GiB -- at the default (0) every
// synchronisation hands a pool's free memory back to the driver and the next call pays for mapping it again (a packed 4096^3 product
// timed 4 calls per synchronise lost 60 us per call to that: profiles/r04/colmajor_a_probe_v1.jsonl against configs_v5.jsonl).  The
// device's DEFAULT pool belongs to the host application and its other libraries: its threshold is never touched.
// Freed with hipFreeAsync like any stream-ordered allocation.  A runtime that cannot make a pool: plain hipMallocAsync.
struct ScratchPools {
  std::atomic<hipMemPool_t> pool[64] = {};
  std::atomic<unsigned long long> failed{0};     // one bit per device ordinal whose pool could not be made
};
inline ScratchPools g_scratch_pools;
inline hipError_t scratch_alloc_async(void **p, size_t bytes, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || (g_scratch_pools.failed.load(std::memory_order_relaxed) >> dev & 1ull))
    return hipMallocAsync(p, bytes, s);
  hipMemPool_t pool = g_scratch_pools.pool[dev].load(std::memory_order_acquire);
  if (pool == nullptr) {
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t made = nullptr;
    if (hipMemPoolCreate(&made, &props) != hipSuccess || made == nullptr) {
      (void)hipGetLastError();
      g_scratch_pools.failed.fetch_or(1ull << dev, std::memory_order_relaxed);
      return hipMallocAsync(p, bytes, s);
    }
    uint64_t keep = (uint64_t)4 << 30;
    (void)hipMemPoolSetAttribute(made, hipMemPoolAttrReleaseThreshold, &keep);
    hipMemPool_t expect = nullptr;
    if (g_scratch_pools.pool[dev].compare_exchange_strong(expect, made, std::memory_order_acq_rel)) {
      pool = made;
    } else {      // another host thread made this device's pool first
      (void)hipMemPoolDestroy(made);
      pool = expect;
    }
  }
  return hipMallocFromPoolAsync(p, bytes, pool, s);
}
// laser_hip_finalize: the pools go (every device is idle by then; what they kept returns to the driver)
inline void scratch_pools_trim() {
  for (int dev = 0; dev < 64; dev++) {
    hipMemPool_t pool = g_scratch_pools.pool[dev].exchange(nullptr, std::memory_order_acq_rel);
    if (pool != nullptr) (void)hipMemPoolDestroy(pool);
  }
  g_scratch_pools.failed.store(0, std::memory_order_relaxed);
}

// One-time-per-DEVICE initialisation of a kernel (hipFuncSetAttribute acts on the current device's instance of the
// function): a single process may drive all 8 GPUs of a node (the sharded entry points), so "done" is a bit per
// device ordinal, not a per-process flag.
struct PerDeviceOnce {
  std::atomic<uint64_t> mask{0};
  template <typename F>
  hipError_t run(F &&f) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = f();
    if (e == hipSuccess) mask.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// One (possibly batched) strided GEMM problem: C <- alpha*A*B + beta*C, element X[r,c] at
// X[r*rs + c*cs] (strides in elements) -- the MatrixView triple of gemm_utils.nim:36-60.
template <typename T>
struct GemmArgs {
  int64_t M, N, K;
  T alpha, beta;
  const T *A;
  int64_t rsA, csA, bsA;
  const T *B;
  int64_t rsB, csB, bsB;
  T *C;
  int64_t rsC, csC, bsC;
  // Readable extents of the operand allocations: rows of A / cols of B / k of both may be read up to
  // these bounds (values beyond M/N/K are zero).  Equal to M, N, K for plain operands; larger for the
  // tile-padded panel images made by the pre-pack API, which lets ragged shapes use the vector loaders.
  int64_t Mext, Next, Kext;
  int32_t tiles_m, tiles_n;  // grid decomposition (filled by the launcher)
  int32_t kc;                // accumulation slice in k (Laser's kc; 0 = one chain over all K)
  int32_t batch;
  int32_t dbg;  // ablation switches for the tuning probe build only (0 in production)
  // Implicit-GEMM convolution (B loader LOAD_IM2COL): B is never materialised; element (k, j) of
  // image b's [C*kH*kW, oH*oW] matrix is gathered from the NCHW input at B + b*bsB with the index
  // arithmetic of im2col (benchmarks/convolution/conv2d_im2col.nim:62-87).
  int32_t cH, cW, ckH, ckW, coW, cpH, cpW, csH, csW;
  // Fused epilogue (the reference plans it: README.md:238-242, TODOs gemm.nim:196,
  // gemm_ukernel_generic.nim:78-79): C = act(alpha*AB + beta*C + bias), bias a strided (broadcastable:
  // strides may be 0) M x N view, applied ONCE after the last accumulation slice.  bias == nullptr / act == 0: off.
  const T *bias;
  int64_t rsBias, csBias, bsBias;
  int32_t act;  // laser_hip_activation
  int32_t cRp, cPWs, cCHM;  // LOAD_CONV_PATCH: patch rows per channel, patch row stride (floats), channels per K-tile
  int32_t cdc, cdr, cdq;  // per-K-tile advance of the implicit-GEMM loader's (channel, kernel row, kernel col): BK = cdc*kH*kW + cdr*kW + cdq
  // First column of C this launch computes: the launch covers columns [col0, N).  Lets one problem be cut into a
  // "main" launch of whole rounds of large tiles (N = the cut) and a "tail" launch of small tiles (col0 = the cut);
  // tiles are independent and every configuration is bit-identical, so the result does not depend on the cut.
  int64_t col0;
  // Implicit-GEMM conv, K-slice-parallel form (tail launches of the laser-order conv): batch = cs_imgs images x nsl kc slices,
  // workgroup z = blockIdx.y computes image z % cs_imgs over k in [slice*cs_len, min(K, (slice+1)*cs_len)), slice = z / cs_imgs,
  // as ONE chain into C + z*bsC (a workspace the ordered combine pass folds); cs_len is a multiple of every BK.  0 = off.
  int32_t cs_imgs, cs_len;
  // Completion flags (small-matrix kernel on host-mapped operands only; nullptr = off): workgroup w stores done_seq to
  // done_flags[w] (system scope, after its C stores) so the host-pointer entry point can poll mapped memory instead of
  // paying a stream synchronise.
  uint32_t *done_flags;
  uint32_t done_seq;
  // Fused prologue (README.md:243-244: "fuse operations before the matrix multiplication kernel, during the prepacking"): relu
  // (x > 0 ? x : 0) applied to the elements of A / of B as they are read.  Only run_gemm and the assembly launcher look at these:
  // run_gemm either hands the problem to the `_pre` assembly kernels (the operation happens in the staging registers, on the way
  // into the LDS panel image) or materialises the operand once and clears the flag.
  int32_t preA, preB;
};

// How an operand tile is brought from HBM into its LDS panel image (the GPU analogue of
// pack_A_mc_kc / pack_B_kc_nc, gemm_packing.nim:24-94 -- strides are resolved HERE).
enum LoadMode : int {
  LOAD_VEC_X = 0,  // unit stride along the tile's M/N dimension, 16-B vector loads, full tiles
  LOAD_VEC_K = 1,  // unit stride along k, 16-B vector loads + transposing LDS write, full tiles
  LOAD_GEN_X = 2,  // any strides / ragged edges, scalar predicated loads, lanes run along M/N
  LOAD_GEN_K = 3,  // any strides / ragged edges, scalar predicated loads, lanes run along k
  // 16-B vector loads on ragged problems (extents multiples of 4, not of the tile): addresses are
  // clamped into the operand, quads beyond K are replaced by zeros (Laser zero-pads its panels the
  // same way, gemm_packing.nim:46-55,85-94; rows/cols beyond M/N only feed outputs never stored)
  LOAD_VEC_X_EDGE = 4,
  LOAD_VEC_K_EDGE = 5,
  // B operand only: im2col fused into the loader (implicit-GEMM convolution), lanes run along
  // the output pixel index j = oh*oW + ow, predicated gathers from the NCHW image, zeros for padding
  LOAD_IM2COL = 6,
  LOAD_CONV_PATCH = 7,  // implicit-GEMM conv, B from an LDS-resident input patch: the tile's input rows are loaded once
                        // (contiguous 16-B pieces) and the kH*kW shifted views are gathered by the fragment reads
};

// LOAD_CONV_PATCH geometry for a BK x BN tile: channels a K-tile can touch, input rows reached by BN consecutive output
// pixels (upper bound), and the patch row stride in floats (input column j at index 4 + j).  The row stride is the
// smallest multiple of 4 >= W + 8 that keeps the 32 lanes of a fragment read on 32 distinct LDS banks when their 32
// consecutive output pixels wrap from one output row to the next: the wrap moves the address by PWs*sH - (oW-1)*sW,
// which must be = sW (mod 32).  (With W + 8 = 64 at the 56x56 shape every patch row starts on bank 0 and the wrapped
// lanes collide 2-way: SQ_LDS_BANK_CONFLICT = 13 % of the LDS cycles, profiles/r02/rocprof_conv_c4.)  Returns false if
// no stride fits the B region of an LDS stage (budget = BK*BN - BK - 4 floats).
struct ConvPatchGeom {
  int chm, rp, pws;
};
inline bool conv_patch_geom(int BK, int BN, int W, int oW, int kH, int kW, int sH, int sW, ConvPatchGeom *g) {
  const int khw = kH * kW;
  g->chm = (BK + khw - 2) / khw + 1;
  g->rp = ((BN - 1) / oW + 1) * sH + kH;
  const int64_t budget = (int64_t)BK * BN - BK - 4;
  g->pws = W + 8;
  if ((int64_t)g->chm * g->rp * g->pws > budget) return false;
  for (int p = W + 8; p <= W + 40; p += 4)
    if ((p * sH - oW * sW) % 32 == 0 && (int64_t)g->chm * g->rp * p <= budget) {
      g->pws = p;
      break;
    }
  return true;
}

hipError_t launch_gemm_f32(const GemmArgs<float> &args, int cfg, bool laser_order, hipStream_t s);
hipError_t launch_gemm_f64(const GemmArgs<double> &args, bool laser_order, hipStream_t s);
// float32, unit column strides on A and C, B row-major-like or passed transposed, any alpha / beta / K, enough tiles to fill the
// chip: the hand-scheduled assembly kernels (gemm_f32_asm.cpp, laser_amd/asmgen/); hipErrorNotSupported = not that class, use launch_gemm_f32
hipError_t launch_gemm_f32_asm(const GemmArgs<float> &args, bool laser_order, hipStream_t s);
// implicit-GEMM convolution on the same framework (3x3 kernel, stride 1, padding 0 or 1): output pixels [0, args.N) of every image
hipError_t launch_conv_f32_asm(const GemmArgs<float> &args, bool laser_order, hipStream_t s);
int64_t conv_asm_plan_cut(const GemmArgs<float> &args, bool laser_order);   // pixel cut for an assembly main launch (-1: not its class, 0: none)
int asm_plan_f32(int64_t M, int64_t N, int64_t K, int laser_order, int cus, int64_t out[8]);   // diagnostics: kernel + launch plan for a device of `cus` CUs (no device touched)
const char *asm_error_detail();   // the assembly launcher's explanation of the error it just returned on this thread ("" = none)
int64_t asm_fixup_timeouts();  // diagnostics: fix-ups of cut launches that gave up waiting (0 in a correct run; synchronises the device)
void asm_kernels_release();   // unload the assembly kernels' code objects (laser_hip_finalize)
hipError_t launch_gemm_f64_asm(const GemmArgs<double> &args, bool laser_order, hipStream_t s);
extern std::atomic<int> g_f64_asm, g_last_f64_asm;
hipError_t launch_gemm_i32_asm(const GemmArgs<int32_t> &args, void *ws, hipStream_t s);
hipError_t launch_gemm_i64_asm(const GemmArgs<int64_t> &args, void *ws, hipStream_t s);
extern std::atomic<int> g_i32_asm, g_last_i32_asm, g_int_group_m;
extern std::atomic<int> g_asm_tile;   // option "asm_tile" (gemm_f32_asm.cpp)
void asm_set_thread_tile(int tile_class);   // per-thread pin of the same (-2 = none)
int asm_get_thread_tile();
int asm_tile_pin_now();                     // the pin this thread's launches see (-1 = none)
extern std::atomic<int> g_last_asm_group_m;
extern std::atomic<int> g_last_asm_rem;       // tiles the last assembly launch left to the K-cut launch of a hybrid plan (0: one launch)
extern std::atomic<int> g_asm_plan, g_asm_kernel, g_asm_wgs, g_asm_slice, g_asm_noseed, g_asm_group_m, g_asm_giveup;   // launch-plan overrides of the assembly kernels (tuning sweeps, tests)
extern std::atomic<int> g_last_asm_wgs, g_last_asm_slices;                  // diagnostics: workgroups / K slices per tile of the last assembly launch
extern std::atomic<int> g_f32_asm;       // 1 default; 0 = never; 2 = whenever the kernel can (no tile-count rule: tests)
extern std::atomic<int> g_last_f32_asm;  // 0 = the last f32 GEMM launch was a compiler-scheduled kernel, 1 / 2 = laser-order / fast assembly kernel
// args.B = NCHW input, args.bsB = C*H*W, args.c* = geometry, N = oH*oW, K = C*kH*kW; A = filter
hipError_t launch_conv_implicit_f32(const GemmArgs<float> &args, int cfg, bool laser_order, hipStream_t s);
int gemm_f32_config_count();
const char *gemm_f32_config_name(int cfg);

// matrix-vector-like problems (M <= 8 or N <= 8) as an HBM stream; hipErrorNotSupported = not skinny, use the tiled kernels
template <typename T>
hipError_t launch_gemm_skinny(const GemmArgs<T> &args, bool laser_order, int kc_elems, hipStream_t s);
// small-matrix path (gemm_small.hip): one wave per 32x32 (f64: 16x16) block of C, operands loaded straight into the
// MFMA operand registers; hipErrorNotSupported = not a small problem, use the tiled kernels.  float32 / float64.
template <typename T>
hipError_t launch_gemm_small(const GemmArgs<T> &args, bool laser_order, int kc_elems, hipStream_t s, bool mapped = false);
extern std::atomic<int> g_small_path;
// the dispatch rule of launch_gemm_small (mapped: the operands live in host memory mapped into the device)
bool gemm_small_takes(int elem_size, int64_t M, int64_t N, int64_t K, int64_t batch, bool mapped = false);
template <typename T>
hipError_t launch_gemm_valu(const GemmArgs<T> &args, bool laser_order, hipStream_t s);

// int32 GEMM on the int8 matrix cores (signed 8-bit limb decomposition, bit-exact mod 2^32);
// ws = device scratch of gemm_i32_mfma_workspace_bytes(M, N, K) bytes, valid on stream s.
size_t gemm_i32_mfma_workspace_bytes(int64_t M, int64_t N, int64_t K);
hipError_t launch_gemm_i32_mfma(const GemmArgs<int32_t> &args, void *ws, hipStream_t s);
// int64 GEMM on the int8 matrix cores (eight signed 8-bit limbs, 36 limb products, bit-exact mod 2^64; K in chunks of
// 8192 per launch); ws = device scratch of gemm_i64_mfma_workspace_bytes(M, N, K) bytes, valid on stream s.
size_t gemm_i64_mfma_workspace_bytes(int64_t M, int64_t N, int64_t K);
hipError_t launch_gemm_i64_mfma(const GemmArgs<int64_t> &args, void *ws, hipStream_t s);

extern std::atomic<int> g_conv_direct;        // few output channels x short K: the direct (HBM-streaming) kernel (1, default)
hipError_t launch_conv_direct_small_f32(const GemmArgs<float> &a, hipStream_t s);
extern std::atomic<int> g_conv_patch;         // implicit conv: LDS input patch where it fits (1, default) or always the gather (0)
extern std::atomic<int> g_last_conv_tail;
extern std::atomic<int> g_conv_walk;          // assembly conv main launch as unit walkers with pipelined transitions: 1 where units > slots (default), 0 never, 2 always, >= 3 that many workgroups (tests)
extern std::atomic<int> g_conv_cut_always;    // tests / probes: cut every 3x3 convolution at its last whole 128-pixel tile (0, default)
extern std::atomic<int> g_conv_tail;          // the direct tail kernel behind the assembly conv main launch (1, default)
hipError_t launch_conv_tail_f32(const GemmArgs<float> &a, int kc, hipStream_t s);
extern std::atomic<int> g_conv_kslice;        // laser-order conv tail as parallel kc slices + ordered combine (1, default)
extern std::atomic<int> g_split_tail;        // 1 (default): cut problems with a badly filled last round into main + tail launches
extern std::atomic<int64_t> g_last_split;    // diagnostics: column cut of the last MFMA launch (0 = one launch)
extern std::atomic<int> g_last_f32_cfg;       // diagnostics: the f32 tile configuration the last GEMM / conv launch used
hipError_t launch_transpose_batched(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC,
                                    int elem_size, hipStream_t s);
hipError_t launch_transpose_pitched(void *dst, int64_t ld_dst, const void *src, int64_t ld_src, int64_t NR, int64_t NC, int elem_size,
                                    hipStream_t s);
extern std::atomic<int> g_im2col_band;
hipError_t launch_im2col(void *ws, int64_t oH, int64_t oW, const void *in, int64_t batch, int64_t C, int64_t H, int64_t W, int64_t kH,
                         int64_t kW, int64_t pH, int64_t pW, int64_t sH, int64_t sW, int elem_size, hipStream_t s);
hipError_t launch_im2col_f32(float *ws, int64_t oH, int64_t oW, const float *in, int64_t batch,
                             int64_t C, int64_t H, int64_t W, int64_t kH, int64_t kW, int64_t pH,
                             int64_t pW, int64_t sH, int64_t sW, hipStream_t s);
// dst[r*ld + c] = (r < R && c < Ccols) ? src[r*rs + c*cs] : 0 for r < Rpad, c < Cpad
// ordered combine of per-kc-slice partial products W[nsl][M][N] into C (slice-parallel GEMM)
template <typename E>
hipError_t launch_combine_slices(E *C, int64_t rsC, int64_t csC, const E *W, int64_t M, int64_t N, int nsl, E alpha, E beta,
                                 hipStream_t s);
// rank-N strided element copy (tensor deepCopy / copyFrom); LASER_MAXRANK = 6 (laser/dynamic_stack_arrays.nim:6)
constexpr int kMaxRank = 6;
template <typename T>
hipError_t launch_copy_strided(T *dst, const int64_t *dstrides, const T *src, const int64_t *sstrides,
                               const int64_t *shape, int rank, hipStream_t s);
// elementwise map over strided rank <= 6 views (map_strided.hip): dst = f(a [, b]); nin = operands read (0: fill)
template <typename T>
hipError_t launch_map_strided(int op, int nin, T *dst, const int64_t *dstrides, const T *a, const int64_t *astrides, const T *b,
                              const int64_t *bstrides, const int64_t *shape, int rank, T alpha, T beta, hipStream_t s);
template <typename T>
hipError_t launch_pack_pad(T *dst, int64_t Rpad, int64_t Cpad, const T *src, int64_t R,
                           int64_t Ccols, int64_t rs, int64_t cs, hipStream_t s, int relu = 0);

}  // namespace laser_hip
