// laser_amd/csrc/gemm_f32_asm.cpp -- host side of the hand-scheduled fp32 GEMM kernels (gfx950 assembly emitted by
// laser_amd/asmgen/f32_kernel.py, assembled at build time, carried in this library as a code object and loaded with
// hipModuleLoadData).  One workgroup = one tile of C, 4 waves = one wave per SIMD with the accumulators in AGPRs -- the
// register-level design of the reference's generated micro-kernels (gemm_ukernel_generator.nim:140-250) with the loop
// nest of gemm.nim:109-176 around it.  The workgroup -> tile map (the ic / jr partition of gemm.nim:160-176) is arithmetic in the
// kernel on a few numbers made here (XCD-aware chunking of the workgroup ids, grouped raster: neighbouring workgroups of an XCD
// share operand panels in its L2), and a launch may be persistent: fewer workgroups than tiles, tiles cut at K-slice boundaries.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include <cstddef>
#include <cstring>
#include "common.h"
#include "limb_planes.h"
#include "f32_asm_blob.h"  // generated: lh_f32_asm_hsaco[], lh_f32_asm_hsaco_len

namespace laser_hip {

std::atomic<int> g_int_group_m{4};   // tile rows per raster group of the integer limb kernels (option int_group_m; 8192^3 int32: 4 is 2-3 % ahead of 8, profiles/r06/i8_raster*_a{r,s}.jsonl)
std::atomic<int> g_i32_asm{1};       // int32 / int64 limb GEMMs: the hand-scheduled kernels when eligible (0 = the compiler-scheduled ones)
std::atomic<int> g_last_i32_asm{0};
std::atomic<int> g_f64_asm{1};       // float64: same meaning as g_f32_asm
std::atomic<int> g_last_f64_asm{0};  // diagnostics: 0 = compiler-scheduled kernel; 1 + index into kKernels otherwise
std::atomic<int> g_f32_asm{1};       // 1 (default): eligible problems run on the hand-scheduled kernels
std::atomic<int> g_last_f32_asm{0};  // diagnostics: 0 = compiler-scheduled kernel; 1 + index into kKernels otherwise

std::atomic<int> g_last_asm_rem{0};    // diagnostics: tiles the last launch left to the second launch of a hybrid plan (0: one launch)
std::atomic<int> g_asm_plan{0};       // option "asm_plan": 0 = the launch model decides; 1 = one tile per workgroup only; 2 = the persistent plan with K-slice cuts whenever legal; 3 = the strided whole-tile plan whenever legal; 4 = strided whole rounds + a K-cut launch over the remaining tiles (hybrid) whenever legal
std::atomic<int> g_asm_kernel{-1};    // option "asm_kernel": force an index of kKernels (tuning sweeps); -1 = the model decides
std::atomic<int> g_asm_tile{-1};      // option "asm_tile": pin a tile class of the f32 GEMM kernels (0 = 256x128 / 256x256, 1 = 256x128 one chain, 2 = 128x128x16, 3 = 128x128x32, 4 = 64x64; on 16x16 blocks: 5 = 96x96, 6 = 160x96, 7 = 128x96, 8 = 192x96, 9 = 160x160; -1 = the model decides)
thread_local int tl_asm_tile = -2;    // the same pin for the launches made BY THIS THREAD (-2 = none: the option applies); asm_set_thread_tile
std::atomic<int> g_asm_wgs{0};        // option "asm_wgs": workgroups of a persistent launch (0 = every slot of the chip)
std::atomic<int> g_asm_slice{0};      // option "asm_slice": K-tiles per slice of a cut one-chain launch (0 = the model decides)
std::atomic<int> g_asm_group_m{0};    // option "asm_group_m": tile rows per raster group of the f32 / f64 GEMM launches (0 = 4 for 256-row tiles, else 8)
std::atomic<int> g_asm_noseed{0};     // option "asm_noseed": 1 = a piece never takes its received sum early (tests: forces the two-run receive path)
std::atomic<int> g_asm_giveup{0};     // option "asm_test_giveup": 1 = every receiver of a cut launch gives up at once, as if its ~2 s of polling had run out (tests of the error report)
std::atomic<int> g_last_asm_wgs{0}, g_last_asm_slices{0}, g_last_asm_group_m{0};   // diagnostics: workgroups / K slices per tile of the last assembly launch

void asm_set_thread_tile(int tile_class) { tl_asm_tile = tile_class; }
int asm_get_thread_tile() { return tl_asm_tile; }
int asm_tile_pin_now() { return tl_asm_tile >= -1 ? tl_asm_tile : (int)g_asm_tile; }

namespace {

struct KernelInfo {
  const char *symbol;
  int bm, bn, bk;
  // tile-choice model (round 4: refitted to profiles/r04/plan_sweep_f32_mid_v2.jsonl, every candidate x plan forced; round 6: the 128x128x16 tile's
  // alone-on-its-CU figure 0.935 -> 0.92 -- 2048^3, one round of 256 tiles: 128.4 us against the 32-deep variant's 126.6, x16_ab_mid_j.jsonl): fraction of the matrix peak a CU reaches on this tile when
  // its workgroup slots are full / when one workgroup has the CU to itself, and the launch's fixed cost (prologue, first
  // loads, epilogue of the last round) in microseconds
  double eff, eff_alone, fixed_us;
  int occ;   // workgroups of this kernel a CU holds (registers / LDS)
};
// [0] laser-order, large tile  [1] one chain, large tile  [2] laser-order, 128x128  [3] one chain, 128x128;
// [4..7] the same with B passed transposed (unit ROW stride: k-contiguous like A, BASELINE configs[2])
// [8] / [9]: one chain on the 256x128x32 tile (plain / B transposed): finer tile quantisation for the fast mode
// [10] / [11]: implicit-GEMM convolution, 3x3 kernel, stride 1, any zero padding (laser-order / one chain)
// [12..15]: 64x64 tiles, three workgroups per CU (laser-order / one chain, plain / B transposed): problems of few tiles
// (1024^3 = 32 tiles of 256x128 for 256 CUs) and the tile quantisation of mid-size ones (3072^3 = 1.125 rounds of 256x128)
// [16..19]: float64 (v_mfma_f64_16x16x4_f64; laser_amd/asmgen/f64_kernel.py): 128x128x16 laser-order / one chain, 64x64x16 same
// [20]: int32 via int8 limb planes (laser_amd/asmgen/i8_kernel.py)
// [21..24]: convolution with fewer output channels: 128x128x32 (laser-order / one chain), 64x128x32 (same)
// [25..28]: float64 with B passed transposed: 128x128x16 (laser-order / one chain), 64x64x16 (same)
// [29]: int64 via eight int8 limb planes (i8_kernel.py "i64_64x64x32")
// [30..33]: 128x128 tiles with a 32-deep K-tile, one workgroup per CU (laser-order / one chain, plain / B transposed): one round of
// 129 .. 256 tiles, where a workgroup has its CU to itself
// [34..45]: fused prologue (relu on A's and / or B's elements in the staging registers): `_pre` variants of [0] [4] [1] [5] [2] [6] [3]
// [7] [12] [14] [13] [15], one tile per workgroup
// [46..53]: float32 on 16x16 blocks (v_mfma_f32_16x16x4_f32; laser_amd/asmgen/f32x16_kernel.py): tiles whose sides are multiples of 32
// -- 96x96 (laser-order / one chain, plain / B transposed) and 160x96 (same) -- for the problems the 32x32-block tiles quantise badly:
// the reference's own benchmark shape 1920^3 (gemm_bench_float32.nim:383-410) is 240 tiles of 160x96 against 225 of 128x128 on 256
// CUs; 1536^3 is 256 tiles of 96x96 against 144 of 128x128
// [54..65]: the same family's 128x96, 192x96 and 160x160 tiles (four variants each, in that order)
// [66..71]: the convolution kernels [10] [11] [21..24] as unit walkers (f32_kernel.py Cfg.cpers): a workgroup runs units (image, tile)
// g, g + G, g + 2G ... and goes from one to the next inside its K loop (the next unit's first gathers before the old tile's last
// K-tiles are multiplied, the old tile's C stores from the gaps of the new tile's first K-tile body)
constexpr int kNumKernels = 72;
const KernelInfo kKernels[kNumKernels] = {
    {"lh_f32_exact_256x128x32", 256, 128, 32, 0.965, 0.965, 10.0, 1},    {"lh_f32_fast_256x256x16", 256, 256, 16, 0.98, 0.98, 12.0, 1},
    {"lh_f32_exact_128x128x16", 128, 128, 16, 0.95, 0.92, 6.0, 2},       {"lh_f32_fast_128x128x16", 128, 128, 16, 0.96, 0.93, 6.0, 2},
    {"lh_f32_exact_256x128x32_nt", 256, 128, 32, 0.965, 0.965, 10.0, 1}, {"lh_f32_fast_256x256x16_nt", 256, 256, 16, 0.98, 0.98, 12.0, 1},
    {"lh_f32_exact_128x128x16_nt", 128, 128, 16, 0.95, 0.92, 6.0, 2},    {"lh_f32_fast_128x128x16_nt", 128, 128, 16, 0.96, 0.93, 6.0, 2},
    {"lh_f32_fast_256x128x32", 256, 128, 32, 0.97, 0.97, 10.0, 1},       {"lh_f32_fast_256x128x32_nt", 256, 128, 32, 0.97, 0.97, 10.0, 1},
    {"lh_f32_conv_exact_256x128x32", 256, 128, 32, 0.88, 0.88, 15.0, 1}, {"lh_f32_conv_fast_256x128x32", 256, 128, 32, 0.88, 0.88, 15.0, 1},
    {"lh_f32_exact_64x64x32", 64, 64, 32, 0.90, 0.84, 6.0, 3},           {"lh_f32_fast_64x64x32", 64, 64, 32, 0.91, 0.85, 6.0, 3},
    {"lh_f32_exact_64x64x32_nt", 64, 64, 32, 0.90, 0.84, 6.0, 3},        {"lh_f32_fast_64x64x32_nt", 64, 64, 32, 0.91, 0.85, 6.0, 3},
    {"lh_f64_exact_128x128x16", 128, 128, 16, 0.937, 0.945, 8.0, 1},     {"lh_f64_fast_128x128x16", 128, 128, 16, 0.965, 0.97, 8.0, 1},
    {"lh_f64_exact_64x64x16", 64, 64, 16, 0.915, 0.815, 3.0, 2},         {"lh_f64_fast_64x64x16", 64, 64, 16, 0.93, 0.83, 3.0, 2},
    {"lh_i32_128x128x32", 128, 128, 32, 0.8, 0.8, 10.0, 1},
    {"lh_f32_conv_exact_128x128x32", 128, 128, 32, 0.8, 0.8, 12.0, 1}, {"lh_f32_conv_fast_128x128x32", 128, 128, 32, 0.8, 0.8, 12.0, 1},
    {"lh_f32_conv_exact_64x128x32", 64, 128, 32, 0.72, 0.72, 8.0, 2},  {"lh_f32_conv_fast_64x128x32", 64, 128, 32, 0.72, 0.72, 8.0, 2},
    {"lh_f64_exact_128x128x16_nt", 128, 128, 16, 0.937, 0.945, 8.0, 1},   {"lh_f64_fast_128x128x16_nt", 128, 128, 16, 0.965, 0.97, 8.0, 1},
    {"lh_f64_exact_64x64x16_nt", 64, 64, 16, 0.915, 0.815, 3.0, 2},       {"lh_f64_fast_64x64x16_nt", 64, 64, 16, 0.93, 0.83, 3.0, 2},
    {"lh_i64_64x64x32", 64, 64, 32, 0.7, 0.7, 10.0, 1},
    {"lh_f32_exact_128x128x32", 128, 128, 32, 0.95, 0.95, 7.0, 1},       {"lh_f32_fast_128x128x32", 128, 128, 32, 0.96, 0.96, 7.0, 1},
    {"lh_f32_exact_128x128x32_nt", 128, 128, 32, 0.95, 0.95, 7.0, 1},    {"lh_f32_fast_128x128x32_nt", 128, 128, 32, 0.96, 0.96, 7.0, 1},
    {"lh_f32_exact_256x128x32_pre", 256, 128, 32, 0.92, 0.92, 10.0, 1},  {"lh_f32_exact_256x128x32_pre_nt", 256, 128, 32, 0.92, 0.92, 10.0, 1},
    {"lh_f32_fast_256x256x16_pre", 256, 256, 16, 0.93, 0.93, 12.0, 1},   {"lh_f32_fast_256x256x16_pre_nt", 256, 256, 16, 0.93, 0.93, 12.0, 1},
    {"lh_f32_exact_128x128x16_pre", 128, 128, 16, 0.90, 0.86, 6.0, 2},   {"lh_f32_exact_128x128x16_pre_nt", 128, 128, 16, 0.90, 0.86, 6.0, 2},
    {"lh_f32_fast_128x128x16_pre", 128, 128, 16, 0.91, 0.87, 6.0, 2},    {"lh_f32_fast_128x128x16_pre_nt", 128, 128, 16, 0.91, 0.87, 6.0, 2},
    {"lh_f32_exact_64x64x32_pre", 64, 64, 32, 0.84, 0.74, 3.0, 3},       {"lh_f32_exact_64x64x32_pre_nt", 64, 64, 32, 0.84, 0.74, 3.0, 3},
    {"lh_f32_fast_64x64x32_pre", 64, 64, 32, 0.85, 0.76, 3.0, 3},        {"lh_f32_fast_64x64x32_pre_nt", 64, 64, 32, 0.85, 0.76, 3.0, 3},
    // (fitted to profiles/r06/x16_ab_{ref,mid}_g.jsonl, x16_ab_more_i.jsonl; the laser-order entries + 0.007 with the running sum in VGPRs, x16_ab_big2_n.jsonl; the many-round figures of pipe_ab_x16_p.jsonl at 7680^3 -- 0.963 / 0.972 on 160x160, 0.934 / 0.952 on 96x96 --: plain launches at 1536^3 .. 5120^3, 1000x3000x2000; the 64x64 tiles'
    // fixed cost 3 -> 6 us from the same runs: 1664^3 and 1000x3000x2000 took 84 / 98 us where the table said 80 / 92)
    {"lh_f32x16_exact_96x96x32", 96, 96, 32, 0.925, 0.925, 5.0, 1},        {"lh_f32x16_fast_96x96x32", 96, 96, 32, 0.943, 0.93, 5.0, 1},
    {"lh_f32x16_exact_96x96x32_nt", 96, 96, 32, 0.925, 0.925, 5.0, 1},     {"lh_f32x16_fast_96x96x32_nt", 96, 96, 32, 0.943, 0.93, 5.0, 1},
    {"lh_f32x16_exact_160x96x32", 160, 96, 32, 0.952, 0.958, 6.0, 1},    {"lh_f32x16_fast_160x96x32", 160, 96, 32, 0.96, 0.963, 6.0, 1},
    {"lh_f32x16_exact_160x96x32_nt", 160, 96, 32, 0.952, 0.958, 6.0, 1}, {"lh_f32x16_fast_160x96x32_nt", 160, 96, 32, 0.96, 0.963, 6.0, 1},
    {"lh_f32x16_exact_128x96x32", 128, 96, 32, 0.938, 0.922, 5.5, 1},    {"lh_f32x16_fast_128x96x32", 128, 96, 32, 0.935, 0.94, 5.5, 1},
    {"lh_f32x16_exact_128x96x32_nt", 128, 96, 32, 0.938, 0.922, 5.5, 1}, {"lh_f32x16_fast_128x96x32_nt", 128, 96, 32, 0.935, 0.94, 5.5, 1},
    {"lh_f32x16_exact_192x96x32", 192, 96, 32, 0.95, 0.952, 6.5, 1},    {"lh_f32x16_fast_192x96x32", 192, 96, 32, 0.962, 0.965, 6.5, 1},
    {"lh_f32x16_exact_192x96x32_nt", 192, 96, 32, 0.95, 0.952, 6.5, 1}, {"lh_f32x16_fast_192x96x32_nt", 192, 96, 32, 0.962, 0.965, 6.5, 1},
    {"lh_f32x16_exact_160x160x32", 160, 160, 32, 0.959, 0.945, 8.0, 1},  {"lh_f32x16_fast_160x160x32", 160, 160, 32, 0.968, 0.956, 8.0, 1},
    {"lh_f32x16_exact_160x160x32_nt", 160, 160, 32, 0.959, 0.945, 8.0, 1}, {"lh_f32x16_fast_160x160x32_nt", 160, 160, 32, 0.968, 0.956, 8.0, 1},
    {"lh_f32_conv_exact_256x128x32_p", 256, 128, 32, 0.90, 0.90, 15.0, 1}, {"lh_f32_conv_fast_256x128x32_p", 256, 128, 32, 0.90, 0.90, 15.0, 1},
    {"lh_f32_conv_exact_128x128x32_p", 128, 128, 32, 0.82, 0.82, 12.0, 1}, {"lh_f32_conv_fast_128x128x32_p", 128, 128, 32, 0.82, 0.82, 12.0, 1},
    {"lh_f32_conv_exact_64x128x32_p", 64, 128, 32, 0.74, 0.74, 8.0, 2},    {"lh_f32_conv_fast_64x128x32_p", 64, 128, 32, 0.74, 0.74, 8.0, 2}};
// plain kernel -> its `_pre` variant (-1: none)
int pre_variant(int k) {
  switch (k) {
    case 0: return 34; case 4: return 35; case 1: return 36; case 5: return 37; case 2: return 38; case 6: return 39;
    case 3: return 40; case 7: return 41; case 12: return 42; case 14: return 43; case 13: return 44; case 15: return 45;
    default: return -1;
  }
}
constexpr int kFullCUs = 256;      // the unpartitioned MI355X (SPX): what the efficiency table was measured on
// Kernels whose persistent workgroups go from one whole tile to the next without leaving the K loop (asmgen/f32_kernel.py Cfg.pipe:
// the next tile's first K-tiles are fetched by the last bodies of this one, its first body stores this one's C): what one such
// transition saves against a fresh workgroup per tile, in microseconds (0: the kernel has no pipelined transition).  Measured:
// profiles/r06/pipe_*.jsonl.
double pipe_gain_us(int k) {
  // profiles/r06/pipe_ab_{big,mid}_b.jsonl (plain vs strided, interleaved, same bits): 256x128x32 +0.2 ... +0.5 % at 4096^3 ... 8192^3,
  // +1.6 % at 5120^3, +1.3 ... +2.3 % on the convolution's GEMM twin (8192x3072x1152: three tiles of 36 K-tiles per workgroup)
  if (k == 0 || k == 4 || k == 8 || k == 9) return 2.5;       // 256x128x32 (laser-order / one chain, B plain / transposed): one workgroup per CU
  if (k >= 30 && k <= 33) return 1.0;                         // 128x128x32 (one workgroup per CU): +0.3 ... +3 %, a tile the model rarely picks
  if (k >= 46 && k <= 65) return 1.5;                         // the 16x16-block tiles (one workgroup per CU; f32x16_kernel.py trans_after)
  if (k == 16 || k == 17 || k == 25 || k == 26) return 3.0;   // float64 128x128x16 (one workgroup per CU; f64_kernel.py trans_after)
  // 256x256x16: -1.8 ... +1.1 % (sixteen blocks' stores in the first sixteen gaps of a 16-deep body): left alone.  Two or three
  // workgroups per CU (128x128x16, 64x64) cover each other's transitions already, and a static share of the tiles quantises in
  // workgroup slots where the plain launch quantises in CUs: -0.3 ... -14 %
  return 0.0;
}
// Workspace of the cut launches of ONE stream on one device: partial tiles + their flags (all flags are zero between launches: the
// workgroup that consumes a partial clears its flag).  Launches on a stream run in order, so they can share it.
struct StreamWs {
  void *ws = nullptr;
  size_t ws_bytes = 0;
  uint32_t *flags = nullptr;
  size_t nflags = 0;
  // flags[0] is the stream's error word: a receiver that gave up waiting for its hand-over counts itself there (f32_kernel.py
  // recv_block).  Behind every cut launch the word is read back -- an asynchronous 4-byte copy on the launch stream into this pinned
  // host word, no synchronisation -- and the NEXT launch on the stream that finds it non-zero fails (check_stream_poison).
  volatile uint32_t *host_err = nullptr;
};
struct DeviceModule {
  std::mutex mu;     // module load + this device's workspace map; never held across a launch
  hipModule_t mod = nullptr;
  hipFunction_t fn[kNumKernels] = {};
  std::map<hipStream_t, StreamWs> ws;
  std::vector<void *> retired;   // outgrown workspaces / flag arrays: freed by laser_hip_finalize only (get_ws)
  int cus = 0;                   // compute units of the device (the launch plans are written for the full 256-CU, 8-XCD MI355X)
};
constexpr int kMaxDev = 16;
DeviceModule g_mods[kMaxDev];
constexpr size_t kWsMaxBytes = (size_t)512 << 20;
constexpr size_t kWsMaxStreams = 64;

// the scheduler block of the kernel arguments (f32_kernel.py KA_SCHED): the workgroup -> tile map is arithmetic in the kernel
// (XCD-aware chunking of the workgroup ids + grouped raster: the ic / jr partition of gemm.nim:160-176), no table
struct SchedArgs {
  uint32_t tiles_m, tiles_n, group_m, gsz_last, mg_width, mg_gm, mg_last, xcd_q;
  uint32_t xcd_r, P, mg_P, units_q, units_r, slice_len, flags_bits, mg_G;
  uint64_t ws, flags;
};
static_assert(sizeof(SchedArgs) == 80, "f32_kernel.py KA_SCHED");

struct KernArgs {
  const void *A, *B;   // float or double
  void *C;
  const uint32_t *unused_;
  uint32_t lda, ldb, ldc, M, N, K;
  float alpha, beta;   // C = beta * C + alpha * A B (beta == 0: C is never read)
  void *dbg;
  // convolution kernels only (f32_kernel.py KA_CONV*)
  uint32_t H, W, oW, pH, pW, Cin, Npix, magic_oW;
  uint32_t shift_oW, pad_;
  uint64_t bsB_bytes, bsC_bytes;
  // fused epilogue of the f32 kernels (f32_kernel.py KA_BIAS / KA_EPI): bias view + activation; zero = plain epilogue
  const void *bias = nullptr;
  uint32_t rsBias = 0, csBias = 0, act = 0, csC = 0;   // csC: column stride of C in elements (f32 GEMM kernels; 0 = 1)
  SchedArgs sch;
};
static_assert(sizeof(KernArgs) == 232 && offsetof(KernArgs, sch) == 152, "kernel argument block layout (f32_kernel.py KA_*)");

// x / d in the kernels = mulhi(x, magic(d)) with magic(d) = floor(2^32 / d) + 1 (0 for d == 1): exact while x * d < 2^32
inline uint32_t magic_u32(uint64_t d) { return d <= 1 ? 0u : (uint32_t)(((uint64_t)1 << 32) / d + 1); }

// Tile map + unit arithmetic of a launch of G workgroups over tiles_m x tiles_n tiles, each cut into P slices of slice_len
// elements of K (P == 1: no cut).  two_level (launches that cut tiles, G a multiple of 8): XCD x = workgroup id % 8 owns the whole
// tiles [T x / 8, T (x + 1) / 8) and its G / 8 workgroups share them -- a cut tile's sender is always started before its receiver
// (asmgen/f32_kernel.py KA_SCHED).  false: a quotient of the in-kernel arithmetic would leave the range of its magic number.
bool fill_sched(SchedArgs &sc, int64_t tiles_m, int64_t tiles_n, int group_m, int64_t G, int64_t P, int64_t slice_len,
                const StreamWs *w, bool xcd, bool two_level, int64_t T_cover = 0) {
  // (T_cover: the launch covers that many tiles of the raster only -- the unit arithmetic runs over them, the raster constants stay
  // those of the whole grid; the kernel adds the launch's first tile, KA_TAB)
  const int64_t Tgrid = tiles_m * tiles_n, T = T_cover > 0 ? T_cover : Tgrid, U = T * P;
  if (group_m <= 0 || group_m > tiles_m) group_m = (int)tiles_m;   // one group: tile rows fastest (the convolution's order)
  const int64_t width = (int64_t)group_m * tiles_n, gsz_last = tiles_m % group_m ? tiles_m % group_m : group_m;
  if ((double)Tgrid * (double)width >= 4.0e9 || (double)U * (double)P >= 4.0e9 || G < 1 || G > U) return false;
  const int64_t q = U / G, r = U % G;
  if ((double)G * (double)r * (double)G >= 4.0e9 || tiles_m > 0x7fffffff || tiles_n > 0x7fffffff || U > 0x7fffffff || Tgrid >= ((int64_t)1 << 28)) return false;
  if (two_level && (G % 8 != 0 || T < 8 || (T / 8) * P < G / 8 || (double)(U / 8 + P) * (double)(G / 8) >= 4.0e9)) return false;
  sc.tiles_m = (uint32_t)tiles_m; sc.tiles_n = (uint32_t)tiles_n; sc.group_m = (uint32_t)group_m; sc.gsz_last = (uint32_t)gsz_last;
  sc.mg_width = magic_u32((uint64_t)width); sc.mg_gm = magic_u32((uint64_t)group_m); sc.mg_last = magic_u32((uint64_t)gsz_last);
  sc.xcd_q = (xcd || two_level) && G >= 8 ? (uint32_t)(G / 8) : 0; sc.xcd_r = (xcd || two_level) && G >= 8 ? (uint32_t)(G % 8) : 0;
  sc.P = (uint32_t)P; sc.mg_P = magic_u32((uint64_t)P); sc.units_q = (uint32_t)(two_level ? T : q); sc.units_r = (uint32_t)(two_level ? 0 : r);
  sc.slice_len = (uint32_t)slice_len; sc.flags_bits = (g_asm_noseed ? 1u : 0u) | (two_level ? 2u : 0u) | (g_asm_giveup ? 8u : 0u);
  sc.mg_G = magic_u32((uint64_t)(two_level ? G / 8 : G));
  sc.ws = w ? (uint64_t)(uintptr_t)w->ws : 0; sc.flags = w ? (uint64_t)(uintptr_t)(w->flags + 1) : 0;   // (flags[0] = the error word)
  return true;
}

hipError_t get_module(int dev, DeviceModule **out) {
  if (dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
  DeviceModule &m = g_mods[dev];
  std::lock_guard<std::mutex> lk(m.mu);
  if (m.cus == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
    m.cus = n;
  }
  // Round 6 (VERDICT r5 missing #4): the launch plans take the device's own CU count (device_cus) -- rounds, workgroup slots and the
  // tile-count thresholds scale with it -- so a partitioned MI355X (CPX 32 / QPX 64 / DPX 128 CUs) keeps the hand-scheduled kernels.
  // Nothing in the kernels depends on the number of XCDs for correctness: the id remap (g % 8 chunks) is a bijection for any grid, and
  // the hand-over order of the cut plans only needs workgroup g to be started before g + 8, which the dispatcher's id order gives.
  if (m.cus < 8) return hipErrorNotSupported;
  if (!m.mod) {
    hipError_t e = hipModuleLoadData(&m.mod, lh_f32_asm_hsaco);
    if (e != hipSuccess) return e;
    for (int k = 0; k < kNumKernels && e == hipSuccess; k++) e = hipModuleGetFunction(&m.fn[k], m.mod, kKernels[k].symbol);
    if (e != hipSuccess) {
      (void)hipModuleUnload(m.mod);
      m.mod = nullptr;
      return e;
    }
  }
  *out = &m;
  return hipSuccess;
}

// compute units of the current device (cached by get_module); kFullCUs when it cannot be asked (the callers fail later, loudly)
int current_cus() {
  int dev = 0;
  DeviceModule *m = nullptr;
  if (hipGetDevice(&dev) != hipSuccess || get_module(dev, &m) != hipSuccess) return kFullCUs;
  return m->cus;
}

// The stream's workspace, grown when needed (rare: an allocation + a clear on the launch stream).  Addresses handed out stay valid
// until laser_hip_finalize: a buffer that is outgrown is RETIRED, not freed -- a launch another host thread queued on the same stream
// a moment ago, or one still running, keeps reading it; growth is geometric, so the retired buffers of a stream add up to
// less than its current one.  Never while the stream is being captured (checked FIRST, also when a buffer is cached): a captured cut
// launch would bake the workspace and its flags into a graph that may be replayed on another stream, or beside eager launches on this
// one, sharing hand-over slots -- hipErrorNotSupported, and the caller takes the plan that needs no workspace.
hipError_t get_ws(DeviceModule *m, hipStream_t s, size_t ws_bytes, size_t nflags, StreamWs *out) {
  if (ws_bytes > kWsMaxBytes) return hipErrorNotSupported;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return hipErrorNotSupported;
  std::lock_guard<std::mutex> lk(m->mu);
  auto it = m->ws.find(s);
  if (it != m->ws.end() && it->second.ws_bytes >= ws_bytes && it->second.nflags >= nflags) {
    *out = it->second;
    return hipSuccess;
  }
  if (it == m->ws.end() && m->ws.size() >= kWsMaxStreams) return hipErrorNotSupported;
  StreamWs w = it != m->ws.end() ? it->second : StreamWs();
  hipError_t e = hipSuccess;
  if (w.ws_bytes < ws_bytes) {
    const size_t want = std::min(kWsMaxBytes, std::max({ws_bytes, 2 * w.ws_bytes, (size_t)32 << 20}));
    void *nw = nullptr;
    e = hipMalloc(&nw, want);
    if (e == hipSuccess) {
      if (w.ws) m->retired.push_back(w.ws);
      w.ws = nw; w.ws_bytes = want;
    }
  }
  if (e == hipSuccess && w.nflags < nflags) {
    const size_t want = std::max({nflags, 2 * w.nflags, (size_t)1 << 16});
    uint32_t *nf = nullptr;
    e = hipMalloc((void **)&nf, want * sizeof(uint32_t));
    // cleared ON THE LAUNCH STREAM: a null-stream hipMemset may still be pending when it returns, and a non-blocking stream's
    // kernel is not ordered behind it -- the clear would then wipe flags the running launch has set (seen: a receiver timing out
    // on the first cut launch of a fresh stream, profiles/r04/README.md)
    if (e == hipSuccess) e = hipMemsetAsync(nf, 0, want * sizeof(uint32_t), s);
    if (e == hipSuccess && !w.host_err) {
      void *he = nullptr;
      e = hipHostMalloc(&he, 64, hipHostMallocDefault);
      if (e == hipSuccess) {
        w.host_err = (volatile uint32_t *)he;
        *w.host_err = 0;
      }
    }
    if (e == hipSuccess) {
      if (w.flags) m->retired.push_back(w.flags);
      w.flags = nf; w.nflags = want;
    } else if (nf) {
      (void)hipFree(nf);
    }
  }
  m->ws[s] = w;
  if (e != hipSuccess) return e;
  *out = w;
  return hipSuccess;
}

thread_local char tl_asm_err[256] = "";

// A cut launch on stream `s` of this device reported receivers that gave up (their tiles hold wrong sums): fail THIS call -- the
// reference aborts on a violated precondition (gemm_prepacked.nim:125), it never returns a wrong C -- say which stream, and make the
// stream usable again: the error word and every hand-over flag are cleared on the stream (a sender that was only late sets its flag
// after its receiver has gone; the next cut launch would take that stale sum), ordered before whatever is launched next.
hipError_t check_stream_poison(DeviceModule *m, hipStream_t s) {
  std::lock_guard<std::mutex> lk(m->mu);
  auto it = m->ws.find(s);
  if (it == m->ws.end() || !it->second.host_err) return hipSuccess;
  const uint32_t n = *it->second.host_err;
  if (n == 0) return hipSuccess;
  *it->second.host_err = 0;
  (void)hipMemsetAsync(it->second.flags, 0, it->second.nflags * sizeof(uint32_t), s);
  snprintf(tl_asm_err, sizeof tl_asm_err,
           "an earlier K-cut launch on stream %p: %u workgroup(s) gave up waiting for a running sum -- the result of that launch is invalid "
           "(the stream's hand-over flags have been reset; later calls are clean)", (void *)s, n);
  return hipErrorLaunchFailure;
}

// One launch of the tiled problem described by (tiles, K): which plan.
//   plain:      one tile per workgroup, the hardware hands tiles to CUs as they free up: ceil(T / slots) rounds.
//   persistent: G = every slot of the chip (or fewer), each workgroup walks an equal share of the T * P units (unit = one K slice of
//               one tile; laser-order: slice = kc, the unit Laser's own pc loop restarts its chain at, gemm.nim:150-158; one chain: a
//               slice length chosen here).  A tile that straddles two workgroups is started by the owner of its slice 0, which sends
//               the running sum on through the workspace (one tile-sized hand-over per cut, whatever P); the next workgroup
//               continues it, in order: laser-order results are the SAME bits as the sequential loop's (asmgen/f32_kernel.py sched_next).
struct Plan {
  bool persistent = false;
  bool strided = false;      // persistent, whole tiles only: workgroup v takes tiles v, v + G, v + 2G ... (pipelined transitions)
  int64_t G = 0, P = 1, slice_len = 0;
  double time_us = 1e300;
  // hybrid (round 6; strided only): the strided launch covers the whole rounds of the raster, tiles [0, T - rem_tiles); the remaining
  // rem_tiles tiles -- a fraction of a round that would cost every workgroup's wait for the few that got one more tile -- follow as a
  // second launch of rem_G workgroups that cuts them along K into rem_P slices (the kernels' tile base, KA_TAB; f32_kernel.py prologue)
  int64_t rem_tiles = 0, rem_G = 0, rem_P = 1, rem_slice_len = 0;
};

Plan plan_launch(const KernelInfo &ki, int64_t tiles, int64_t K, int64_t batch, bool exact, int kc, double cu_flops_per_us, bool may_cut,
                 double pipe_us = -1.0, int cus = kFullCUs, bool allow_hybrid = true) {
  const int64_t kCUs = cus;
  Plan best;
  const double tile_us = 2.0 * ki.bm * ki.bn * (double)K / cu_flops_per_us;
  {
    const int64_t t = tiles * batch, rounds = (t + kCUs - 1) / kCUs;
    best.G = tiles;
    best.slice_len = (K + ki.bk - 1) / ki.bk * ki.bk;
    best.time_us = (double)rounds * tile_us / (rounds == 1 ? ki.eff_alone : ki.eff) + ki.fixed_us;
    // More tiles than workgroup slots and a kernel that pipelines its tile transitions (pipe_us > 0: the caller has checked that this
    // problem may -- beta == 0, plain epilogue, whole K-tiles): every slot of the chip gets one persistent workgroup that walks the
    // tiles v, v + G, v + 2G ...: the same tiles at the same time as the rounds of the plain launch (an XCD's workgroups on one
    // contiguous chunk of the raster), minus what a fresh workgroup per tile costs -- drain, C store burst, prologue, first loads.
    const int64_t slots = g_asm_wgs > 0 ? std::min<int64_t>(g_asm_wgs, 4096) : (int64_t)kCUs * ki.occ;
    // (pipe_us < 0: this problem may not pipeline; asm_plan = 3 forces the plan wherever it is legal -- sweeps, tests)
    if (pipe_us >= 0.0 && batch == 1 && g_asm_plan != 1 && g_asm_plan != 2 && ((pipe_us > 0.0 && tiles > slots) || (g_asm_plan >= 3 && tiles >= 2))) {
      const int64_t G = std::min(slots, tiles), per_wg = (tiles + G - 1) / G;
      if (per_wg >= 2) {
        best.persistent = true;
        best.strided = true;
        best.G = G;
        best.P = 1;
        best.time_us -= (double)(per_wg - 1) * pipe_us;
        best.time_us *= 1.01;      // (a static share of a ragged number of rounds: 5632^3 .. 7936^3 ran 0.5 .. 2 % over this estimate, x16_ab_big2_n.jsonl)
      }
    }
    if (g_asm_plan == 3) return best;
  }
  if (g_asm_plan == 1 || !may_cut || batch != 1 || !g_split_tail) return best;
  const int64_t slots = g_asm_wgs > 0 ? std::min<int64_t>(g_asm_wgs, 4096) : (int64_t)kCUs * ki.occ;
  const int64_t kt = (K + ki.bk - 1) / ki.bk;
  // the best plan that cuts `tiles` tiles along K (remainder: the tiles are what a strided launch leaves over -- ranges shorter than a
  // tile are what it is for)
  const auto best_cut = [&](int64_t tiles, bool remainder) {
  Plan pers;
  // candidate cuts: laser-order: the kc slices; one chain: 1..16 slices of equal numbers of K-tiles (or the forced length)
  int64_t cuts[18];
  int ncuts = 0;
  if (exact) {
    cuts[ncuts++] = kc;
  } else if (g_asm_slice > 0) {
    cuts[ncuts++] = (int64_t)g_asm_slice * ki.bk;
  } else {
    for (int64_t p = 1; p <= 16 && p <= kt; p++) {
      // (two or three long slices per tile: every workgroup's range is a send piece + a receive piece and the launch runs about one
      // slice longer than the model says -- 3328^3 one chain on 256x128 in 3 slices: 117.9 TFLOP/s where the plain 160x96 launch runs
      // 137.7; 2048^3 / 1920^3 in 2: 0.77 / 0.69 of peak, profiles/r06/size_sweep_vendor_j.jsonl, x16_ab_mid_g.jsonl)
      if ((p == 2 || p == 3) && g_asm_plan != 2 && !remainder) continue;
      const int64_t len = (kt + p - 1) / p * ki.bk;
      if (len >= 4 * ki.bk || p == 1) cuts[ncuts++] = len;
    }
  }
  for (int i = 0; i < ncuts; i++) {
    const int64_t len = cuts[i], P = (K + len - 1) / len, U = tiles * P;
    int64_t G = std::min(slots, U);
    if (G >= 8) {                                         // launches that cut tiles: 8 XCDs x G / 8 workgroups (fill_sched), and no
      if (tiles < 8) continue;                            // more workgroups on an XCD than the XCD with the fewest tiles has units
      G = 8 * std::min(G / 8, (tiles / 8) * P);
    }
    // some tile straddles two workgroups (an XCD's share of the tiles may be one more or less than T / 8: then its units do not
    // divide evenly even where U / G does)
    const bool cut = U % G != 0 || (U / G) % P != 0 || (G >= 8 && tiles % 8 != 0);
    if (!cut && G >= tiles && g_asm_plan != 2 && !remainder) continue;  // one whole tile per workgroup: that is the plain launch
    // many tiles per workgroup: the workgroups of an XCD drift apart in the tile order and lose their shared panels -- the plain
    // launch keeps them on neighbouring tiles
    if ((double)U / (double)G / (double)P > 4.0 && g_asm_plan != 2) continue;
    const int64_t q = U / G, r = U % G;
    // ranges much shorter than a tile: a workgroup waits for the sum of a predecessor that is still computing it (hand-over
    // chains); such problems are few-tile x long-K: the slice-parallel form / the plain launch serve them
    if (q < P - 2 && g_asm_plan != 2 && !remainder) continue;
    // the busiest CU: its workgroups' units back to back (the r longer ranges are spread evenly over the workgroup ids), at the
    // average unit length (the last slice of a tile is shorter)
    const int64_t wg_per_cu = (G + kCUs - 1) / kCUs;
    const double extras = r ? std::min((double)wg_per_cu, std::ceil((double)r * (double)wg_per_cu / (double)G)) : 0.0;
    const double units_cu = (double)(q * wg_per_cu) + extras;
    const double unit_us = tile_us / (double)P;
    // fitted to the round-4 sweeps (profiles/r04/plan_sweep_*).  The 64x64 kernels' rate differs by up to 10 % between boxes
    // (3072^3 plain: 130 TFLOP/s on one MI355X, 145.6 on another in the same hour; the large tiles repeat within 1 %), so their table
    // entries sit at the low end and ties go to the larger tile.  Three-per-CU tiles: a workgroup that walks about one tile keeps
    // its XCD's workgroups on neighbouring tiles and runs a little better than the plain launch's rounds; walking several tiles
    // they drift apart in the tile order and lose shared panels.  A launch that cuts tiles pays ~8 us (the two extra runs of a cut
    // tile, its hand-over through the workspace) whatever its length -- all workgroups pay it at the same time, so the CU's other
    // workgroups do not hide it -- and ranges shorter than a tile minus one slice wait on their predecessors
    const double tiles_per_wg = (double)U / (double)G / (double)P;
    double eff = G >= (int64_t)kCUs * ki.occ ? (ki.occ >= 3 ? (tiles_per_wg <= 1.3 ? std::min(ki.eff + 0.025, 0.97) : ki.eff - 0.045) : ki.eff)
                                             : ki.eff_alone + (ki.eff - ki.eff_alone) * std::min(1.0, (double)(wg_per_cu - 1) / std::max(1, ki.occ - 1));
    double t_us = units_cu * unit_us / eff + ki.fixed_us + (cut ? 8.0 : 0.0);
    if (cut && q < P - 1) t_us *= 1.0 + 0.25 * (double)(P - 1 - q) / (double)P;
    // (round 6: against the plain launches of the 16x16-block tiles -- whose estimates land within 1 % -- the persistent plans ran
    // 3 .. 7 % over this estimate at 2560^3 .. 5120^3 (profiles/r06/x16_ab_*.jsonl: 5120^3 1915-1966 us for 1829, 3584^3 669 for 640,
    // 3072^3 427 for 414, 2560^3 251 for 238): the workgroups of a CU do not stay in step for the whole launch)
    t_us *= exact ? 1.045 : 1.10;      // (one chain: 4 .. 7 % more again -- 3328^3 566 us for 529, 6912^3 4760 for 4513, 7936^3 7120 for 6824: x16_ab_big2_n.jsonl)
    if (t_us < pers.time_us) {       // the best cut; it replaces the plain launch only with a margin (below)
      pers.persistent = true;
      pers.G = G; pers.P = P; pers.slice_len = len;
      pers.time_us = t_us;
    }
  }
  return pers;
  };
  // hybrid: the whole rounds strided, the rest of the raster cut along K over every slot (a strided launch of 5.3 rounds takes 6).
  // Measured (profiles/r06/plan_ab_frac_hybrid_ab.jsonl, 256x128 and 256x256 tiles at 4608^3 .. 7424^3): the second launch takes
  // 0.35 + 0.95 x its share of a round, in tile times -- its workgroups sit on K ranges of their own, so nothing of an operand panel is
  // shared through L2 the way a round of whole tiles shares it -- in ONE-CHAIN mode: 0.46 / 0.62 / 0.71 tile times for 0.13 / 0.28 /
  // 0.53 of a round (6656^3 131 -> 143 TFLOP/s, 5888^3 135 -> 141, 4608^3 137 -> 142).  Laser-order: a workgroup folds its slices onto
  // the received sum IN ORDER, so all but the first slice of its piece wait for the predecessor's whole piece (f32_kernel.py
  // sched_next): with ranges much shorter than a tile the pieces of a tile run one after the other -- 0.76 .. 1.08 tile times whatever
  // the share -- and the extra strided round is no worse: not offered (asm_plan = 4 forces it: same bits).
  if (allow_hybrid && best.strided && g_asm_plan != 3 && tiles % best.G >= 8 && tiles / best.G >= 1 && (!exact || g_asm_plan == 4)) {
    const int64_t R = tiles % best.G, full = tiles / best.G;
    const Plan rem = best_cut(R, true);
    if (rem.persistent) {
      const double t = ((double)full + 0.35 + 0.95 * (double)R / (double)best.G) * tile_us / ki.eff + ki.fixed_us + 8.0 - (double)(full - 1) * std::max(0.0, pipe_us);
      if (t < 0.985 * best.time_us || g_asm_plan == 4) {
        best.rem_tiles = R; best.rem_G = rem.G; best.rem_P = rem.P; best.rem_slice_len = rem.slice_len;
        best.time_us = t;
      }
    }
  }
  if (g_asm_plan == 4) return best;
  const Plan pers = best_cut(tiles, false);
  if (pers.persistent && (pers.time_us < 1.0 * best.time_us || g_asm_plan == 2)) return pers;      // (0.96 before the 1.045 above: the same margin)
  return best;
}

// Fill the scheduler block for `plan` and launch.  The workspace is taken only by launches that cut tiles.
hipError_t launch_planned(DeviceModule *m, int kern, const Plan &plan_in, KernArgs &ka, int tiles_m, int tiles_n, int group_m, int64_t batch,
                          size_t tile_bytes, hipStream_t s, int64_t walk_G = 0) {
  Plan plan = plan_in;
  StreamWs w;
  {
    const hipError_t pe = check_stream_poison(m, s);
    if (pe != hipSuccess) return pe;
  }
  const int64_t T = (int64_t)tiles_m * tiles_n, U = T * plan.P;
  if (plan.strided && plan.rem_tiles > 0) {
    // hybrid: whole rounds strided, then the rest of the raster cut along K (Plan::rem_*).  Everything the second launch needs is
    // prepared BEFORE the first is issued (its workspace is refused while the stream is being captured): if anything is missing the
    // strided launch simply covers all tiles.
    const int64_t R = plan.rem_tiles, T1 = T - R, G2 = plan.rem_G, P2 = plan.rem_P, U2 = R * P2;
    bool ok = walk_G == 0 && batch == 1 && T1 > 0 && T1 % plan.G == 0 && G2 >= 1 && G2 <= U2 && ka.unused_ == nullptr;
    const bool cuts2 = ok && (U2 % G2 != 0 || (U2 / G2) % P2 != 0 || (G2 >= 8 && R % 8 != 0));
    const bool two2 = cuts2 && G2 >= 8 && G2 % 8 == 0 && R >= 8;
    if (cuts2 && G2 >= 8 && !two2) ok = false;
    StreamWs w2;
    if (ok && cuts2) ok = get_ws(m, s, (size_t)G2 * tile_bytes, (size_t)G2 + 1, &w2) == hipSuccess;
    KernArgs ka2 = ka;
    if (ok) ok = fill_sched(ka2.sch, tiles_m, tiles_n, group_m, G2, P2, plan.rem_slice_len, cuts2 ? &w2 : nullptr, group_m > 0 && (!cuts2 || two2), two2, R);
    if (ok) ok = fill_sched(ka.sch, tiles_m, tiles_n, group_m, plan.G, 1, plan.slice_len, nullptr, group_m > 0, false, T1);
    if (ok) {
      ka.sch.units_q = (uint32_t)plan.G;
      ka.sch.units_r = (uint32_t)T1;
      ka.sch.flags_bits |= 4u;
      ka2.unused_ = (const uint32_t *)(uintptr_t)T1;       // KA_TAB's low word: the second launch's first tile
      size_t sz = sizeof(ka);
      void *extra1[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      hipError_t e = hipModuleLaunchKernel(m->fn[kern], (unsigned)plan.G, 1, 1, 256, 1, 1, 0, s, nullptr, extra1);
      if (e != hipSuccess) return e;
      void *extra2[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka2, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      e = hipModuleLaunchKernel(m->fn[kern], (unsigned)G2, 1, 1, 256, 1, 1, 0, s, nullptr, extra2);
      if (e == hipSuccess && cuts2 && w2.host_err) (void)hipMemcpyAsync((void *)w2.host_err, w2.flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) {
        g_last_asm_wgs = (int)plan.G;
        g_last_asm_slices = (int)P2;
        g_last_asm_rem = (int)R;
        g_last_asm_group_m = (int)ka.sch.group_m | (ka.sch.xcd_q ? 1 << 16 : 0);
      }
      return e;
    }
    plan.rem_tiles = 0;
  }
  g_last_asm_rem = 0;
  const bool cuts = plan.persistent && !plan.strided && (U % plan.G != 0 || (U / plan.G) % plan.P != 0 || (plan.G >= 8 && T % 8 != 0));
  const bool two_level = cuts && plan.G >= 8 && plan.G % 8 == 0 && T >= 8;
  if (cuts && plan.G >= 8 && !two_level) return hipErrorNotSupported;
  if (cuts) {
    const hipError_t e = get_ws(m, s, (size_t)plan.G * tile_bytes, (size_t)plan.G + 1, &w);   // one slot per sender; flags[0] = the error word
    if (e == hipErrorNotSupported) return e;
    if (e != hipSuccess) return e;
  }
  if (!fill_sched(ka.sch, tiles_m, tiles_n, group_m, plan.G, plan.P, plan.slice_len, cuts ? &w : nullptr, group_m > 0 && (!cuts || two_level), two_level))
    return hipErrorNotSupported;
  if (plan.strided) {      // workgroup v walks the whole tiles v, v + G, v + 2G ... (f32_kernel.py sched_init: flags bit 2, +44 = stride, +48 = tiles)
    if (cuts || plan.P != 1 || T > 0x7fffffff) return hipErrorNotSupported;
    ka.sch.units_q = (uint32_t)plan.G;
    ka.sch.units_r = (uint32_t)T;
    ka.sch.flags_bits |= 4u;
  }
  unsigned gx = (unsigned)plan.G, gy = (unsigned)batch;
  if (walk_G > 0) {
    // unit walkers (convolution, f32_kernel.py next_unit): units = images x tiles of one image, workgroup g runs g, g + walk_G, ...;
    // +4 the tiles of an image, +8 their magic number (unit / tiles: exact while units * tiles < 2^32), +12 the stride, +16 the units
    const int64_t units = T * batch;
    if (plan.persistent || plan.P != 1 || walk_G > units || (double)units * (double)T >= 4.0e9) return hipErrorNotSupported;
    ka.sch.P = (uint32_t)T;
    ka.sch.mg_P = magic_u32((uint64_t)T);
    ka.sch.units_q = (uint32_t)walk_G;
    ka.sch.units_r = (uint32_t)units;
    gx = (unsigned)walk_G;
    gy = 1;
  }
  size_t sz = sizeof(ka);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(m->fn[kern], gx, gy, 1, 256, 1, 1, 0, s, nullptr, extra);
  // (the error word travels back behind the launch; whoever launches next on this stream looks at it -- no wait here)
  if (e == hipSuccess && cuts && w.host_err) (void)hipMemcpyAsync((void *)w.host_err, w.flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    g_last_asm_wgs = walk_G > 0 ? (int)walk_G : (int)plan.G;
    g_last_asm_slices = (int)plan.P;
    g_last_asm_group_m = (int)ka.sch.group_m | (ka.sch.xcd_q ? 1 << 16 : 0);      // raster group height; bit 16: XCD-aware chunking of the ids
  }
  return e;
}

void zero_conv_fields(KernArgs &ka) {
  ka.unused_ = nullptr;
  ka.dbg = nullptr;
  ka.H = ka.W = ka.oW = ka.pH = ka.pW = ka.Cin = ka.Npix = ka.magic_oW = ka.shift_oW = ka.pad_ = 0;
  ka.bsB_bytes = ka.bsC_bytes = 0;
}

}  // namespace

// what the assembly launcher has to say about the error it just returned on this thread ("" if nothing); cleared by the read
const char *asm_error_detail() {
  static thread_local char out[256];
  snprintf(out, sizeof out, "%s", tl_asm_err);
  tl_asm_err[0] = 0;
  return out;
}

// diagnostics (option "asm_fixup_timeouts", synchronises the device): streams of the current device on which some workgroup gave up
// waiting for a running sum (asmgen/f32_kernel.py recv_block: after ~2 s of polling it stores 1 into the stream's error word and
// goes on with what the slot holds -- a flagged wrong result instead of a hung GPU; never in a correct run).  A sender that was only
// LATE sets its flag after its receiver has given up and cleared it: the flag would still be set when the next cut launch on that
// stream starts and hand it a stale sum.  So a stream found with its error word set gets its whole flag array cleared here (the
// device is idle at this point) -- the launches after the report are clean again.
int64_t asm_fixup_timeouts() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return -1;
  DeviceModule &m = g_mods[dev];
  std::lock_guard<std::mutex> lk(m.mu);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  int64_t n = 0;
  for (auto &kv : m.ws) {
    uint32_t wv = 0;
    if (kv.second.flags && hipMemcpy(&wv, kv.second.flags, 4, hipMemcpyDeviceToHost) == hipSuccess && wv != 0) {
      n += wv;
      (void)hipMemset(kv.second.flags, 0, kv.second.nflags * sizeof(uint32_t));
    }
  }
  if (n) (void)hipDeviceSynchronize();
  return n;
}

// laser_hip_finalize: unload the code objects and free the workspaces of every device that used them
void asm_kernels_release() {
  int cur = -1;
  (void)hipGetDevice(&cur);
  for (int d = 0; d < kMaxDev; d++) {
    DeviceModule &m = g_mods[d];
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.mod) continue;
    (void)hipSetDevice(d);
    (void)hipDeviceSynchronize();
    for (auto &kv : m.ws) {
      if (kv.second.ws) (void)hipFree(kv.second.ws);
      if (kv.second.flags) (void)hipFree(kv.second.flags);
      if (kv.second.host_err) (void)hipHostFree((void *)kv.second.host_err);
    }
    m.ws.clear();
    for (void *p : m.retired) (void)hipFree(p);
    m.retired.clear();
    (void)hipModuleUnload(m.mod);
    m.mod = nullptr;
  }
  if (cur >= 0) (void)hipSetDevice(cur);
}

namespace {
// hipErrorNotSupported: not this kernel's class of problem.  A with unit column stride, B row-major-like or passed transposed, C with
// any positive strides (the epilogue's address arithmetic takes the column stride; rows are the tile's fast direction).
// which kernel and which launch plan for this problem on a device of `cus` compute units; no device is touched (the CPU suite calls
// it through laser_hip_plan_f32)
hipError_t choose_gemm_f32_asm(const GemmArgs<float> &a, bool laser_order, int cus, int *pick_out, Plan *plan_out) {
  if (!g_f32_asm) return hipErrorNotSupported;
  if (a.batch < 1 || a.batch > 65535 || a.col0 != 0 || a.done_flags != nullptr) return hipErrorNotSupported;
  if (a.batch > 1 && (a.bsA < 0 || a.bsB < 0 || a.bsC < 0)) return hipErrorNotSupported;
  // fused epilogue (laser_hip_gemm_strided_ex_*): bias views with non-negative element strides, no activation or relu; tanh /
  // sigmoid stay on the compiler-scheduled kernels (their library bodies are what the bit-exact tests pin)
  const bool fused = a.bias != nullptr || a.act != 0;
  if (a.act != 0 && a.act != 1) return hipErrorNotSupported;
  if (a.bias != nullptr && (a.rsBias < 0 || a.csBias < 0 || a.rsBias > 0x3fffffff || a.csBias > 0x3fffffff || (a.batch > 1 && a.bsBias != 0) ||
                            ((double)(a.M - 1) * a.rsBias + (double)(a.N - 1) * a.csBias + 1.0) * 4.0 >= 2147483648.0))
    return hipErrorNotSupported;
  // C: rows are the slow direction of the view -- a tile's rows beyond M are dropped by the descriptor's size alone (they lie past
  // the last row), which needs every row to end before the next one starts
  if (a.csA != 1 || a.csC < 1 || a.csC > 0x3fffffff || a.rsC < (a.N - 1) * a.csC + 1) return hipErrorNotSupported;
  // (tile-padded pre-pack images -- Mext / Next / Kext beyond M / N / K, gemm_prepacked.nim:63-292 -- are plain padded row-major
  // copies: the kernels bound every access by M, N, K themselves and never need the padding)
  if (a.Mext < a.M || a.Next < a.N || a.Kext < a.K) return hipErrorNotSupported;
  // B: row-major-like (unit column stride), or passed transposed (unit row stride: every column is k-contiguous)
  const bool nt = a.csB != 1 && a.rsB == 1;
  if (!nt && a.csB != 1) return hipErrorNotSupported;
  const int64_t ldb = nt ? a.csB : a.rsB;
  if (a.rsA < a.K || ldb < (nt ? a.K : a.N)) return hipErrorNotSupported;
  // (a ragged last K-tile is zero-filled piece-wise -- 16 bytes = 4 k -- and, when K % 4 != 0, element-wise in the staging registers)
  if (a.K < 1 || a.M < 1 || a.N < 1) return hipErrorNotSupported;
  // laser-order results need the kc = 512 slices only when K > 512; one chain otherwise (the laser-order kernels are
  // plain single-chain kernels then: their fold tile is never reached)
  const bool exact = laser_order && a.K > 512;
  const int big = (exact ? 0 : (a.K > 512 ? 1 : 0)) + (nt ? 4 : 0), small = (exact ? 2 : (a.K > 512 ? 3 : 2)) + (nt ? 4 : 0);
  // 32-bit byte offsets inside the descriptors
  if ((double)a.rsA * 4.0 * 256 >= 4.0e9) return hipErrorNotSupported;
  if ((nt ? (double)ldb * 4.0 * 256 : (double)a.K * (double)ldb * 4.0) >= 4.0e9) return hipErrorNotSupported;
  if (((double)(a.M - 1) * (double)a.rsC + (double)(a.N - 1) * (double)a.csC + 1.0) * 4.0 > 2147483648.0) return hipErrorNotSupported;
  // (batched problems -- gemm_strided_batched, the kc slices of the slice-parallel form -- are grid y: every tile count below is
  // per launch)
  // Which tile and which plan.  Workgroups that share a CU share its matrix pipes, so whatever the number of workgroup slots a
  // plain launch of T tiles takes ceil(T / 256) times one tile's matrix time: the smaller the tile the finer the quantisation, the
  // larger the tile the closer a CU gets to the peak (the eff columns of kKernels); the persistent plan (plan_launch) cuts the
  // quantisation to one K slice.  3072^3: 288 tiles of 256x128 = 2 rounds for 1.125 rounds of work, or 6.75 slices per CU.
  int pick = -1;
  Plan plan;
  const int mid = (!exact && a.K > 512) ? (nt ? 9 : 8) : -1;   // one chain over a long K: also the 256x128 tile
  const int tiny = 12 + ((exact || a.K <= 512) ? 0 : 1) + (nt ? 2 : 0);
  const int deep = 30 + ((exact || a.K <= 512) ? 0 : 1) + (nt ? 2 : 0);    // 128x128 with the 32-deep K-tile: one workgroup per CU
  const double cu_flops_per_us = 157.3e6 / 256.0;
  // the one-chain kernels' fused epilogue has no C read: beta != 0 with a bias / activation only on the laser-order kernels
  const auto lo_kernel = [](int k) { return k == 0 || k == 2 || k == 4 || k == 6 || k == 12 || k == 14 || k == 30 || k == 32 || (k >= 46 && k <= 65 && (k - 46) % 2 == 0); };
  const bool pre = a.preA != 0 || a.preB != 0;
  // A pinned tile class (option "asm_tile" / the sharded entry point's LASER_HIP_SHARD_PIN_TILE on its worker threads): the local
  // products of a multi-GPU run that shares the CUs with RCCL's kernels.  One tile per workgroup then -- a persistent plan counts on
  // every workgroup slot of the chip -- and no lower bound on the tile count (the caller asked for THIS kernel family).
  const int tile_pin = asm_tile_pin_now();
  // 16x16-block tiles (f32x16_kernel.py): 16-byte pieces are all-or-nothing (K % 4 == 0), no fused prologue (the fused epilogue --
  // bias view + relu -- and a column stride on C are theirs too)
  const bool x16_ok = a.K % 4 == 0 && !pre;
  const int x96 = x16_ok ? 46 + ((exact || a.K <= 512) ? 0 : 1) + (nt ? 2 : 0) : -1, x160 = x16_ok ? x96 + 4 : -1;
  const int x128 = x16_ok ? x96 + 8 : -1, x192 = x16_ok ? x96 + 12 : -1, x160s = x16_ok ? x96 + 16 : -1;
  const int classes[10] = {big, mid, small, deep, tiny, x96, x160, x128, x192, x160s};      // (index = the tile class of option "asm_tile")
  // near ties go to the class asked first: within a block family, the larger tile (less L2 traffic, fewer workgroups)
  const int order[10] = {0, 1, 2, 3, 4, 9, 8, 6, 7, 5};
  for (int oi = 0; oi < 10; oi++) {
    const int ci = order[oi];
    const int k0 = classes[ci];
    if (tile_pin >= 0 && ci != (tile_pin == 1 && mid < 0 ? 0 : tile_pin)) continue;
    if (k0 < 0 || (g_asm_kernel >= 0 && k0 != g_asm_kernel)) continue;
    if (fused && !lo_kernel(k0) && a.beta != 0.0f) continue;
    const int k = pre ? pre_variant(k0) : k0;       // fused prologue: the variants that apply it in the staging registers
    if (k < 0) continue;
    const KernelInfo &ki_ = kKernels[k];
    const int64_t tm = (a.M + ki_.bm - 1) / ki_.bm, tn = (a.N + ki_.bn - 1) / ki_.bn, t = tm * tn;
    if ((double)t * 8.0 * (double)tn >= 4.0e9) continue;    // the in-kernel tile arithmetic's range (fill_sched)
    // below ~5/8 of a round of the larger tiles (3/8 of the 64x64 ones) the compiler-scheduled kernels' slice-parallel and
    // small-problem forms do better
    if (g_f32_asm < 2 && tile_pin < 0 && t * a.batch < (k0 == tiny ? 3 : 5) * (int64_t)cus / 8) continue;
    // (laser-order with K <= kc is ONE chain that must stay one chain: cuts only at kc boundaries, or anywhere in one-chain mode;
    // the `_pre` variants run one tile per workgroup)
    const bool may_pipe = !fused && a.beta == 0.0f && a.K % ki_.bk == 0 && a.K >= 3 * ki_.bk && tile_pin < 0 && !pre;
    const Plan p = plan_launch(ki_, t, a.K, a.batch, exact, 512, cu_flops_per_us, (exact || !laser_order) && !pre && tile_pin < 0,
                               may_pipe ? pipe_gain_us(k) : -1.0, cus);
    if (p.time_us < 0.99 * plan.time_us) plan = p, pick = k;   // (near ties go to the larger tile: less L2 traffic)
  }
  if (pick < 0) return hipErrorNotSupported;
  *pick_out = pick;
  *plan_out = plan;
  return hipSuccess;
}

hipError_t launch_gemm_f32_asm_core(const GemmArgs<float> &a, bool laser_order, hipStream_t s) {
  if (!g_f32_asm) return hipErrorNotSupported;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  DeviceModule *m = nullptr;
  e = get_module(dev, &m);
  if (e != hipSuccess) return e;
  int pick = -1;
  Plan plan;
  e = choose_gemm_f32_asm(a, laser_order, m->cus, &pick, &plan);
  if (e != hipSuccess) return e;
  const KernelInfo &ki = kKernels[pick];
  const int tiles_m = (int)((a.M + ki.bm - 1) / ki.bm), tiles_n = (int)((a.N + ki.bn - 1) / ki.bn);
  const bool nt = a.csB != 1 && a.rsB == 1;
  const int64_t ldb = nt ? a.csB : a.rsB;
  const bool exact = laser_order && a.K > 512;
  const double cu_flops_per_us = 157.3e6 / 256.0;
  const int group_m = g_asm_group_m > 0 ? (int)g_asm_group_m : (ki.bm >= 2 * ki.bn) ? 4 : 8;
  KernArgs ka;
  zero_conv_fields(ka);
  ka.A = a.A;
  ka.B = a.B;
  ka.C = a.C;
  ka.lda = (uint32_t)a.rsA;
  ka.ldb = (uint32_t)ldb;
  ka.ldc = (uint32_t)a.rsC;
  ka.M = (uint32_t)a.M;
  ka.N = (uint32_t)a.N;
  ka.K = (uint32_t)a.K;
  ka.alpha = a.alpha;
  ka.beta = a.beta;
  // batch strides in bytes (f32_kernel.py KA_BSA = the H, W slots; B's and C's in the convolution kernels' slots)
  const uint64_t bsA_bytes = a.batch > 1 ? (uint64_t)a.bsA * 4 : 0;
  ka.H = (uint32_t)bsA_bytes;
  ka.W = (uint32_t)(bsA_bytes >> 32);
  ka.bsB_bytes = a.batch > 1 ? (uint64_t)a.bsB * 4 : 0;
  ka.bsC_bytes = a.batch > 1 ? (uint64_t)a.bsC * 4 : 0;
  ka.bias = a.bias;
  ka.rsBias = a.bias ? (uint32_t)a.rsBias : 0;
  ka.csBias = a.bias ? (uint32_t)a.csBias : 0;
  ka.act = (uint32_t)a.act;
  ka.csC = a.csC == 1 ? 0u : (uint32_t)a.csC;
  ka.pad_ = (a.preA ? 1u : 0u) | (a.preB ? 2u : 0u);      // f32_kernel.py KA_PRE (read by the `_pre` variants only)
  e = launch_planned(m, pick, plan, ka, tiles_m, tiles_n, group_m, a.batch, (size_t)ki.bm * ki.bn * 4, s);
  if (e == hipErrorNotSupported && plan.persistent) {   // no workspace (a stream being captured, ...): one tile per workgroup
    Plan plain = plan_launch(ki, (int64_t)tiles_m * tiles_n, a.K, a.batch, exact, 512, cu_flops_per_us, false, -1.0, m->cus);
    e = launch_planned(m, pick, plain, ka, tiles_m, tiles_n, group_m, a.batch, (size_t)ki.bm * ki.bn * 4, s);
  }
  if (e == hipSuccess) {
    g_last_f32_asm = 1 + pick;
    g_last_split = 0;  // one launch
  }
  return e;
}
}  // namespace

// laser_hip_plan_f32 (diagnostics; touches no device): the kernel and launch plan the f32 launcher would take for a dense row-major
// M x N x K product on a device of `cus` compute units.  out[0] = 1 + kernel index (0: the compiler-scheduled kernels take it),
// [1] = plan (0 one tile per workgroup / 1 persistent with K-slice cuts / 2 strided whole tiles), [2] = workgroups, [3] = K slices
// per tile, [4] = tiles, [5] = tile rows, [6] = tile columns, [7] = workgroup slots of the device for this kernel.
int asm_plan_f32(int64_t M, int64_t N, int64_t K, int laser_order, int cus, int64_t out[8]) {
  for (int i = 0; i < 8; i++) out[i] = 0;
  if (M < 1 || N < 1 || K < 1 || cus < 8 || cus > 4096) return 1;
  GemmArgs<float> a;
  std::memset(&a, 0, sizeof a);
  a.M = M; a.N = N; a.K = K; a.alpha = 1.0f; a.beta = 0.0f;
  a.rsA = K; a.csA = 1; a.rsB = N; a.csB = 1; a.rsC = N; a.csC = 1;
  a.Mext = M; a.Next = N; a.Kext = K; a.batch = 1;
  int pick = -1;
  Plan plan;
  if (choose_gemm_f32_asm(a, laser_order != 0, cus, &pick, &plan) != hipSuccess) return 0;
  const KernelInfo &ki = kKernels[pick];
  out[0] = 1 + pick;
  out[1] = plan.strided ? (plan.rem_tiles > 0 ? 3 : 2) : plan.persistent ? 1 : 0;     // (3: hybrid -- [3] then holds the K slices of the second launch)
  if (plan.rem_tiles > 0) plan.P = plan.rem_P;
  out[2] = plan.G;
  out[3] = plan.P;
  out[5] = (M + ki.bm - 1) / ki.bm;
  out[6] = (N + ki.bn - 1) / ki.bn;
  out[4] = out[5] * out[6];
  out[7] = (int64_t)cus * ki.occ;
  return 0;
}

// Any MatrixView (gemm_utils.nim:36-60: element strides on all three operands; README.md:211-213 advertises `myTensor[:, 0::2]`
// and column-major operands) onto the kernels above:
//   * C whose columns are its slow direction (column-major-like, colStride > rowStride): C^T = B^T A^T -- every element is the same
//     k-ascending chain, so the bits are the same -- which turns C into a row-major-like view;
//   * A with a column stride / B with neither stride 1: the operand is packed once into a dense row-major scratch copy (a
//     transposing pass for a unit-row-stride source: 16-byte accesses on both HBM sides; an element gather otherwise) -- Laser
//     packs every panel it touches (gemm_packing.nim:24-94), here only operands the tile loaders cannot stream pay that pass
//     (2 x 67 MB at 4096^2: ~3 % of the product's time);
//   * C with a column stride: the kernels' own epilogue.
// hipErrorNotSupported: not this kernel family's class of problem -- the caller takes the compiler-scheduled kernels.
hipError_t launch_gemm_f32_asm(const GemmArgs<float> &a_in, bool laser_order, hipStream_t s) {
  if (!g_f32_asm) return hipErrorNotSupported;
  GemmArgs<float> a = a_in;
  if (a.csC > a.rsC && a.rsC >= 1 && a.batch == 1) {     // columns are the slow direction of C (column-major, with or without a row stride)
    std::swap(a.M, a.N);
    std::swap(a.Mext, a.Next);
    const float *pa = a.A;
    const int64_t rsa = a.rsA, csa = a.csA;
    a.A = a.B; a.rsA = a.csB; a.csA = a.rsB;
    a.B = pa;  a.rsB = csa;   a.csB = rsa;
    std::swap(a.rsC, a.csC);
    std::swap(a.rsBias, a.csBias);
    std::swap(a.preA, a.preB);
  }
  const bool packA = a.csA != 1, packB = a.csB != 1 && a.rsB != 1;
  if (!packA && !packB) return launch_gemm_f32_asm_core(a, laser_order, s);
  // packing pays for itself only on products that keep the chip busy for a while; plain / fused-epilogue single problems
  if (a.batch != 1 || a.Mext != a.M || a.Next != a.N || a.Kext != a.K || a.col0 != 0 || a.done_flags != nullptr) return hipErrorNotSupported;
  if ((double)a.M * (double)a.N * (double)a.K < 1024.0 * 1024.0 * 1024.0 || a.K < 64) return hipErrorNotSupported;
  if ((packA && (a.rsA < 0 || a.csA < 0)) || (packB && (a.rsB < 0 || a.csB < 0))) return hipErrorNotSupported;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return hipErrorNotSupported;
  const size_t bytesA = packA ? (size_t)a.M * a.K * 4 : 0, bytesB = packB ? (size_t)a.K * a.N * 4 : 0;
  if (bytesA + bytesB > ((size_t)4 << 30)) return hipErrorNotSupported;
  float *scratch = nullptr;
  hipError_t e = scratch_alloc_async((void **)&scratch, bytesA + bytesB, s);
  if (e != hipSuccess) return e;
  if (packA) {      // A[m][k] at m * rsA + k * csA -> dense [M][K]
    // (a fused prologue on a packed operand happens in the packing pass -- "during the prepacking", README.md:243-244)
    // (the transposing pass reads a K x M row-major matrix of pitch csA: a broadcast or overlapping view -- csA < M -- is not one; the
    // element gather takes any non-negative strides)
    e = (a.rsA == 1 && !a.preA && a.csA >= a.M) ? launch_transpose_pitched(scratch, a.K, a.A, a.csA, a.K, a.M, 4, s)
                                                : launch_pack_pad<float>(scratch, a.M, a.K, a.A, a.M, a.K, a.rsA, a.csA, s, a.preA);
    a.A = scratch; a.rsA = a.K; a.csA = 1; a.preA = 0;
  }
  if (e == hipSuccess && packB) {      // neither stride of B is 1: dense [K][N]
    float *sb = scratch + bytesA / 4;
    e = launch_pack_pad<float>(sb, a.K, a.N, a.B, a.K, a.N, a.rsB, a.csB, s, a.preB);
    a.B = sb; a.rsB = a.N; a.csB = 1; a.preB = 0;
  }
  // (a packing pass that refuses its arguments has launched nothing: not this kernel family's problem, the compiler kernels take it)
  if (e == hipErrorInvalidValue) e = hipErrorNotSupported;
  if (e == hipSuccess) e = launch_gemm_f32_asm_core(a, laser_order, s);
  const hipError_t e2 = hipFreeAsync(scratch, s);
  if (e == hipErrorNotSupported) return e;      // (nothing was launched but the packing passes, which touched only the scratch)
  return e != hipSuccess ? e : e2;
}


// int32 GEMM mod 2^32 on the int8 matrix cores (gemm_i32_mfma.hip's arithmetic; kernel of laser_amd/asmgen/i8_kernel.py): the
// packing pass writes tile-major digit planes into `ws` (>= 4 * (rup(M,128) + rup(N,128)) * rup(K,32) bytes), then one launch.
// Any int32 alpha / beta, unit column stride on C, K <= 8192 (the accumulator groups are never folded).
hipError_t launch_gemm_i32_asm(const GemmArgs<int32_t> &a, void *ws, hipStream_t s) {
  if (!g_i32_asm) return hipErrorNotSupported;
  if (a.batch != 1 || a.csC != 1 || a.rsC < a.N) return hipErrorNotSupported;
  if (a.M < 1 || a.N < 1 || a.K < 1 || a.K > 8192) return hipErrorNotSupported;
  const int64_t Mpad = (a.M + 127) / 128 * 128, Npad = (a.N + 127) / 128 * 128, Kpad = (a.K + 31) / 32 * 32;
  const int64_t tiles = (Mpad / 128) * (Npad / 128);
  if (g_i32_asm < 2 && tiles < current_cus() / 2) return hipErrorNotSupported;      // few tiles: the 8-wave compiler kernel's two workgroups per CU
  if (((double)(a.M - 1) * (double)a.rsC + (double)a.N) * 4.0 > 2147483648.0 || (double)tiles * 8.0 * (double)(Npad / 128) >= 4.0e9)
    return hipErrorNotSupported;
  int8_t *Ap = (int8_t *)ws, *Bp = Ap + 4 * Mpad * Kpad;
  hipError_t e = launch_limb_planes<int32_t>(Ap, a.A, a.M, a.K, a.rsA, a.csA, Mpad, Kpad, s, 128);
  if (e != hipSuccess) return e;
  e = launch_limb_planes<int32_t>(Bp, a.B, a.N, a.K, a.csB, a.rsB, Npad, Kpad, s, 128);
  if (e != hipSuccess) return e;
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  DeviceModule *m = nullptr;
  e = get_module(dev, &m);
  if (e != hipSuccess) return e;
  const int tiles_m = (int)(Mpad / 128), tiles_n = (int)(Npad / 128), group_m = g_int_group_m;
  KernArgs ka;
  zero_conv_fields(ka);
  ka.A = Ap; ka.B = Bp; ka.C = a.C;
  ka.lda = (uint32_t)(Kpad / 32); ka.ldb = 0; ka.ldc = (uint32_t)a.rsC;
  ka.M = (uint32_t)a.M; ka.N = (uint32_t)a.N; ka.K = (uint32_t)Kpad;
  static_assert(sizeof(float) == sizeof(int32_t), "");
  std::memcpy(&ka.alpha, &a.alpha, 4);   // int32 alpha / beta travel in the float slots (i8_kernel.py)
  std::memcpy(&ka.beta, &a.beta, 4);
  Plan plain;      // one tile per workgroup (the limb kernels are never cut along K: integer sums need no order, K <= 8192 per launch)
  plain.G = tiles;
  e = launch_planned(m, 20, plain, ka, tiles_m, tiles_n, group_m, 1, 0, s);
  if (e == hipSuccess) g_last_i32_asm = 21;
  return e;
}

// int64 GEMM mod 2^64 (gemm_i64_mfma.hip's arithmetic; kernel "i64_64x64x32" of laser_amd/asmgen/i8_kernel.py): eight tile-major
// digit planes in `ws` (>= 8 * (rup(M,64) + rup(N,64)) * rup(K,32) bytes), one launch.  Any int64 alpha / beta (wrapping), K <= 8192.
hipError_t launch_gemm_i64_asm(const GemmArgs<int64_t> &a, void *ws, hipStream_t s) {
  if (!g_i32_asm) return hipErrorNotSupported;
  if (a.batch != 1 || a.csC != 1 || a.rsC < a.N) return hipErrorNotSupported;
  if (a.M < 1 || a.N < 1 || a.K < 1 || a.K > 8192) return hipErrorNotSupported;
  const int64_t Mpad = (a.M + 63) / 64 * 64, Npad = (a.N + 63) / 64 * 64, Kpad = (a.K + 31) / 32 * 32;
  const int64_t tiles = (Mpad / 64) * (Npad / 64);
  if (g_i32_asm < 2 && tiles < current_cus() / 2) return hipErrorNotSupported;
  if (((double)(a.M - 1) * (double)a.rsC + (double)a.N) * 8.0 > 2147483648.0 || (double)tiles * 8.0 * (double)(Npad / 64) >= 4.0e9)
    return hipErrorNotSupported;
  int8_t *Ap = (int8_t *)ws, *Bp = Ap + 8 * Mpad * Kpad;
  hipError_t e = launch_limb_planes<int64_t>(Ap, a.A, a.M, a.K, a.rsA, a.csA, Mpad, Kpad, s, 64);
  if (e != hipSuccess) return e;
  e = launch_limb_planes<int64_t>(Bp, a.B, a.N, a.K, a.csB, a.rsB, Npad, Kpad, s, 64);
  if (e != hipSuccess) return e;
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  DeviceModule *m = nullptr;
  e = get_module(dev, &m);
  if (e != hipSuccess) return e;
  const int tiles_m = (int)(Mpad / 64), tiles_n = (int)(Npad / 64), group_m = g_int_group_m;
  KernArgs ka;
  zero_conv_fields(ka);
  ka.A = Ap; ka.B = Bp; ka.C = a.C;
  ka.lda = (uint32_t)(Kpad / 32); ka.ldb = 0; ka.ldc = (uint32_t)a.rsC;
  ka.M = (uint32_t)a.M; ka.N = (uint32_t)a.N; ka.K = (uint32_t)Kpad;
  ka.alpha = 0.0f; ka.beta = 0.0f;
  // int64 alpha / beta in the H, W / oW, pH slots (i8_kernel.py KA_ALPHA64 = 72: where the f64 kernels take their doubles)
  static_assert(offsetof(KernArgs, H) == 72 && offsetof(KernArgs, oW) == 80, "KA_ALPHA64");
  std::memcpy(&ka.H, &a.alpha, 8);
  std::memcpy(&ka.oW, &a.beta, 8);
  Plan plain;
  plain.G = tiles;
  e = launch_planned(m, 29, plain, ka, tiles_m, tiles_n, group_m, 1, 0, s);
  if (e == hipSuccess) g_last_i32_asm = 30;
  return e;
}

namespace {
// float64 twin of launch_gemm_f32_asm_core (kernels of laser_amd/asmgen/f64_kernel.py): row-major A and C, B row-major or passed
// transposed, any alpha / beta, K even, batches as grid y.
hipError_t launch_gemm_f64_asm_core(const GemmArgs<double> &a, bool laser_order, hipStream_t s) {
  if (!g_f64_asm) return hipErrorNotSupported;
  if (a.batch < 1 || a.batch > 65535 || a.bias != nullptr || a.act != 0 || a.col0 != 0 || a.done_flags != nullptr) return hipErrorNotSupported;
  if (a.batch > 1 && (a.bsA < 0 || a.bsB < 0 || a.bsC < 0)) return hipErrorNotSupported;
  if (a.csA != 1 || a.csC != 1) return hipErrorNotSupported;
  const bool nt = a.csB != 1 && a.rsB == 1;
  if (!nt && a.csB != 1) return hipErrorNotSupported;
  const int64_t ldb = nt ? a.csB : a.rsB;
  // (tile-padded pre-pack images -- Mext / Next / Kext beyond M / N / K, gemm_prepacked.nim:63-292 -- are plain padded row-major
  // copies: the kernels bound every access by M, N, K themselves and never need the padding)
  if (a.Mext < a.M || a.Next < a.N || a.Kext < a.K) return hipErrorNotSupported;
  if (a.rsA < a.K || ldb < (nt ? a.K : a.N) || a.rsC < a.N || a.K < 2 || a.K % 2 != 0) return hipErrorNotSupported;   // 16-byte pieces = 2 k
  if ((double)a.rsA * 8.0 * 128 >= 4.0e9 || (nt ? (double)ldb * 8.0 * 128 : (double)a.K * (double)ldb * 8.0) >= 4.0e9) return hipErrorNotSupported;
  if (((double)(a.M - 1) * (double)a.rsC + (double)a.N) * 8.0 > 2147483648.0) return hipErrorNotSupported;
  const bool exact = laser_order && a.K > 256;   // kc = 256 doubles (gemm_tiling.nim:310)
  const int big = (nt ? 25 : 16) + (exact ? 0 : 1), tiny = (nt ? 27 : 18) + (exact ? 0 : 1);
  int pick = -1;
  Plan plan;
  const double cu_flops_per_us = 78.6e6 / 256.0;
  const int cus = current_cus();
  for (int k : {big, tiny}) {
    if (g_asm_kernel >= 0 && k != g_asm_kernel) continue;
    const KernelInfo &ki_ = kKernels[k];
    const int64_t tm = (a.M + ki_.bm - 1) / ki_.bm, tn = (a.N + ki_.bn - 1) / ki_.bn, t = tm * tn;   // (batches are grid y)
    if ((double)t * 8.0 * (double)tn >= 4.0e9) continue;
    if (g_f64_asm < 2 && t * a.batch < (k == tiny ? 3 : 5) * (int64_t)cus / 8) continue;
    // (pipelined tile transitions, round 6: beta == 0, whole K-tiles, three or more of them -- f64_kernel.py once())
    const bool may_pipe = a.beta == 0.0 && a.K % ki_.bk == 0 && a.K >= 3 * ki_.bk && a.batch == 1;
    // (the hybrid plan is fitted and tested on the f32 kernels only: not offered here)
    const Plan p = plan_launch(ki_, t, a.K, a.batch, exact, 256, cu_flops_per_us, exact || !laser_order, may_pipe ? pipe_gain_us(k) : -1.0, cus, false);
    if (p.time_us < 0.99 * plan.time_us) plan = p, pick = k;
  }
  if (pick < 0) return hipErrorNotSupported;
  const KernelInfo &ki = kKernels[pick];
  const int tiles_m = (int)((a.M + ki.bm - 1) / ki.bm), tiles_n = (int)((a.N + ki.bn - 1) / ki.bn);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  DeviceModule *m = nullptr;
  e = get_module(dev, &m);
  if (e != hipSuccess) return e;
  const int group_m = 8;
  KernArgs ka;
  zero_conv_fields(ka);
  ka.A = a.A; ka.B = a.B; ka.C = a.C;
  ka.lda = (uint32_t)a.rsA; ka.ldb = (uint32_t)ldb; ka.ldc = (uint32_t)a.rsC;
  ka.M = (uint32_t)a.M; ka.N = (uint32_t)a.N; ka.K = (uint32_t)a.K;
  ka.alpha = 1.0f; ka.beta = 0.0f;
  // alpha, beta as float64 in the H..pH slots (f64_kernel.py KA_ALPHA64)
  uint64_t ab[2];
  static_assert(sizeof(double) == 8, "");
  std::memcpy(&ab[0], &a.alpha, 8);
  std::memcpy(&ab[1], &a.beta, 8);
  ka.H = (uint32_t)ab[0]; ka.W = (uint32_t)(ab[0] >> 32);
  ka.oW = (uint32_t)ab[1]; ka.pH = (uint32_t)(ab[1] >> 32);
  // batch strides in bytes (f64_kernel.py KA_BSA64 = 88: the pW, Cin slots; B's and C's where the f32 kernels take theirs)
  static_assert(offsetof(KernArgs, pW) == 88, "KA_BSA64");
  const uint64_t bsA_bytes = a.batch > 1 ? (uint64_t)a.bsA * 8 : 0;
  ka.pW = (uint32_t)bsA_bytes; ka.Cin = (uint32_t)(bsA_bytes >> 32);
  ka.bsB_bytes = a.batch > 1 ? (uint64_t)a.bsB * 8 : 0;
  ka.bsC_bytes = a.batch > 1 ? (uint64_t)a.bsC * 8 : 0;
  e = launch_planned(m, pick, plan, ka, tiles_m, tiles_n, group_m, a.batch, (size_t)ki.bm * ki.bn * 8, s);
  if (e == hipErrorNotSupported && plan.persistent) {
    Plan plain = plan_launch(ki, (int64_t)tiles_m * tiles_n, a.K, a.batch, exact, 256, cu_flops_per_us, false, -1.0, cus);
    e = launch_planned(m, pick, plain, ka, tiles_m, tiles_n, group_m, a.batch, (size_t)ki.bm * ki.bn * 8, s);
  }
  if (e == hipSuccess) g_last_f64_asm = 1 + pick;
  return e;
}
}  // namespace

// float64: a column-major-like C as C^T = B^T A^T and a transposed A (unit row stride) through the packing pass, like
// launch_gemm_f32_asm; other strides (a column stride on C, gathers) stay on the compiler-scheduled kernels.
hipError_t launch_gemm_f64_asm(const GemmArgs<double> &a_in, bool laser_order, hipStream_t s) {
  if (!g_f64_asm) return hipErrorNotSupported;
  GemmArgs<double> a = a_in;
  if (a.csC != 1 && a.rsC == 1 && a.batch == 1 && a.bias == nullptr) {
    std::swap(a.M, a.N);
    std::swap(a.Mext, a.Next);
    const double *pa = a.A;
    const int64_t rsa = a.rsA, csa = a.csA;
    a.A = a.B; a.rsA = a.csB; a.csA = a.rsB;
    a.B = pa;  a.rsB = csa;   a.csB = rsa;
    std::swap(a.rsC, a.csC);
  }
  if (a.csA == 1) return launch_gemm_f64_asm_core(a, laser_order, s);
  if (a.rsA != 1 || a.batch != 1 || a.Mext != a.M || a.Next != a.N || a.Kext != a.K || a.col0 != 0 || a.done_flags != nullptr || a.csA < a.M) return hipErrorNotSupported;
  if ((double)a.M * (double)a.N * (double)a.K < 1024.0 * 1024.0 * 1024.0 || a.K < 64 || a.K % 2 != 0) return hipErrorNotSupported;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return hipErrorNotSupported;
  double *scratch = nullptr;
  hipError_t e = scratch_alloc_async((void **)&scratch, (size_t)a.M * a.K * 8, s);
  if (e != hipSuccess) return e;
  e = launch_transpose_pitched(scratch, a.K, a.A, a.csA, a.K, a.M, 8, s);
  a.A = scratch; a.rsA = a.K; a.csA = 1;
  if (e == hipSuccess) e = launch_gemm_f64_asm_core(a, laser_order, s);
  const hipError_t e2 = hipFreeAsync(scratch, s);
  if (e == hipErrorNotSupported) return e;
  return e != hipSuccess ? e : e2;
}

namespace {
// what the assembly convolution launcher makes of a call: the kernel (kKernels index), the padded K, the geometry and the tile grid
struct ConvClass {
  int pick = -1, tiles_m = 0, tiles_n = 0;
  int64_t kH = 0, kW = 0, sH = 0, sW = 0, taps = 0, oW = 0, oH = 0, npix = 0, Cin = 0, Kp = 0, tiles = 0;
  bool exact = false;
};
// hipErrorNotSupported: not this launcher's class (launch_conv_f32_asm's comment); no device work
hipError_t conv_asm_classify(const GemmArgs<float> &a, bool laser_order, ConvClass &cc) {
  if (!g_f32_asm) return hipErrorNotSupported;
  if (a.col0 != 0 || a.cs_imgs != 0) return hipErrorNotSupported;
  if (a.alpha != 1.0f || a.beta != 0.0f) return hipErrorNotSupported;
  // fused epilogue (laser_hip_conv2d_im2col_ex_f32: per-channel bias, relu) as in the GEMM kernels; tanh / sigmoid: the compiler kernels
  if (a.act != 0 && a.act != 1) return hipErrorNotSupported;
  if (a.bias != nullptr && (a.rsBias < 0 || a.csBias < 0 || a.bsBias != 0 || a.rsBias > 0x3fffffff || a.csBias > 0x3fffffff ||
                            ((double)(a.M - 1) * a.rsBias + (double)(a.N - 1) * a.csBias + 1.0) * 4.0 >= 2147483648.0))
    return hipErrorNotSupported;
  const int64_t kH = a.ckH, kW = a.ckW, sH = a.csH, sW = a.csW, taps = kH * kW;
  if (kH < 1 || kW < 1 || kH > 255 || kW > 255 || taps > 49 || sH < 1 || sW < 1 || sH > 255 || sW > 255) return hipErrorNotSupported;
  if (a.cpH < 0 || a.cpW < 0 || a.cpH > 255 || a.cpW > 255) return hipErrorNotSupported;
  if (a.cH + 2 * a.cpH < kH || a.cW + 2 * a.cpW < kW) return hipErrorNotSupported;
  const int64_t oW = (a.cW + 2 * a.cpW - kW) / sW + 1, oH = (a.cH + 2 * a.cpH - kH) / sH + 1, npix = oH * oW;
  if (oW != a.coW || oW <= 0 || oH <= 0) return hipErrorNotSupported;
  if (a.K % taps != 0 || a.K < taps || a.csC != 1 || a.rsC != npix) return hipErrorNotSupported;
  if (a.N > npix || (a.N != npix && a.N % 128 != 0)) return hipErrorNotSupported;
  if (a.batch < 1 || a.batch > 65535 || a.M > 0xffff * 64ll) return hipErrorNotSupported;
  const int64_t Cin = a.K / taps;
  const int64_t Kp = (a.K + 3) / 4 * 4;          // filter rows as whole 16-byte pieces
  if ((double)Cin * a.cH * a.cW * 4.0 >= 2.0e9 || (double)a.M * npix * 4.0 >= 2.0e9 || (double)Kp * 4.0 * 256 >= 4.0e9 ||
      (double)Kp * (double)taps >= 4.0e9)      // (the in-kernel k / taps: x * d < 2^32)
    return hipErrorNotSupported;
  const bool exact = laser_order && a.K > 512;
  // rows of the tile by the number of output channels: the smallest padded row count, weighted by what each tile reaches (the
  // 256-row tile's tap table ends at 31 taps)
  int pick = -1;
  double best_cost = 1e300;
  for (int base : {10, 21, 23}) {
    if (base == 10 && taps > 31) continue;
    const KernelInfo &kc = kKernels[base];
    const double cost = (double)((a.M + kc.bm - 1) / kc.bm * kc.bm) / kc.eff;
    if (cost < 0.99 * best_cost) best_cost = cost, pick = base;
  }
  pick += (exact || a.K <= 512 ? 0 : 1);
  const KernelInfo &ki = kKernels[pick];
  const int tiles_m = (int)((a.M + ki.bm - 1) / ki.bm), tiles_n = (int)((a.N + ki.bn - 1) / ki.bn);
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  if ((double)tiles * (double)tiles >= 4.0e9) return hipErrorNotSupported;   // the in-kernel tile arithmetic's range (fill_sched)
  if (g_f32_asm < 2 && tiles * a.batch < 5 * (int64_t)current_cus() / 8) return hipErrorNotSupported;
  // a 256-row tile that is mostly padding (few output channels) loses to the compiler-scheduled 128 / 64-row tiles
  if (g_f32_asm < 2 && (double)a.M * (double)a.N < 0.75 * (double)tiles * ki.bm * ki.bn) return hipErrorNotSupported;
  if ((double)npix * (double)oW >= 4.0e9) return hipErrorNotSupported;
  // the 64-row tile (two workgroups per CU, each at half the CU's rate): a CU works through ceil(units / CUs) units' worth of time
  // whatever the count; where that leaves a fifth of the chip idle the compiler-scheduled kernels' smaller tiles are ahead by 6 - 10 %
  // (profiles/r06/conv_m64_ab_w.jsonl: 800 units = 3.125 per CU, 320 units = 1.25 per CU), elsewhere behind by 4 - 25 %
  if (g_f32_asm < 2 && ki.bm == 64) {
    const int64_t cus = current_cus(), units_ = tiles * a.batch, per_cu = (units_ + cus - 1) / cus;
    if ((double)units_ < 0.8 * (double)(per_cu * cus)) return hipErrorNotSupported;
  }
  cc.pick = pick; cc.tiles_m = tiles_m; cc.tiles_n = tiles_n; cc.tiles = tiles;
  cc.kH = kH; cc.kW = kW; cc.sH = sH; cc.sW = sW; cc.taps = taps; cc.oW = oW; cc.oH = oH; cc.npix = npix; cc.Cin = Cin; cc.Kp = Kp;
  cc.exact = exact;
  return hipSuccess;
}
}  // namespace

// Where to cut the pixel axis of a convolution whose main part the assembly kernels would take (round 6: the launch plan of
// gemm_mfma.hip plan_split is written for the compiler-scheduled configurations' tiles; in laser-order mode it cut 3136 pixels at
// 2048 where the 128-pixel assembly tile wants 3072: 109 against 127 TFLOP/s on 5x5 64 -> 128 channels, conv_geometry_ab_w.jsonl).
// Candidates: the whole image, or the first k whole 128-pixel tiles of every image with the rest as a tail launch.  A CU works
// through ceil(units / CUs) units one after the other (two workgroups of the 64-row tile share a CU at half its rate each: the same
// time); the tail is priced as a launch of its own at a fraction of the chip's rate (C4's 64-pixel tail: 14.4 us).
// Returns -1: not the assembly launcher's class (the caller keeps its own plan), 0: one launch, > 0: the cut.
int64_t conv_asm_plan_cut(const GemmArgs<float> &a_in, bool laser_order) {
  GemmArgs<float> a = a_in;
  const int cus = current_cus();
  if (cus < 8) return -1;
  const double cu_rate = 157.3e12 / 256.0;
  const auto main_us = [&](const ConvClass &cc) {
    const KernelInfo &ki = kKernels[cc.pick];
    const int64_t units = cc.tiles * a.batch;
    const double unit_us = 2.0 * ki.bm * ki.bn * (double)cc.Kp / (cu_rate * (ki.eff + 0.02)) * 1e6;
    return 8.0 + (double)((units + cus - 1) / cus) * unit_us;
  };
  double best = 1e300;
  int64_t best_cut = -1;
  ConvClass cc;
  if (conv_asm_classify(a, laser_order, cc) == hipSuccess) best = main_us(cc), best_cut = 0;
  const int64_t npix = a.N, kfull = npix / 128;
  for (int64_t k = kfull; k >= 1 && k >= kfull - 12; k--) {
    if (k * 128 == npix) continue;
    a.N = k * 128;
    if (conv_asm_classify(a, laser_order, cc) != hipSuccess) continue;
    // (the 64-row tile with a short reduction: a cut launch loses 5 - 10 % to one launch of the compiler-scheduled kernels, K = 288 / 864;
    // from K = 1152 on it is 5 - 9 % ahead: profiles/r06/conv_m64_ab_x.jsonl)
    if (kKernels[cc.pick].bm == 64 && cc.Kp < 1024) continue;
    const double tail_flops = 2.0 * (double)a.M * (double)(npix - k * 128) * (double)a.K * (double)a.batch;
    const double t = main_us(cc) + 12.0 + tail_flops / 50.0e6;       // (us: 12 + flops / 50 TFLOP/s)
    if (t < 0.98 * best) best = t, best_cut = k * 128;
  }
  return best_cut;
}

// Implicit-GEMM convolution (conv2d_im2col.nim:102-166 minus the materialised im2col matrix): output pixels [0, a.N) of every
// image, a.N a multiple of the 128-pixel tile or the whole image.  GemmArgs as launch_conv_implicit_f32 builds them (A = the
// filter [M][K], B = the NCHW input, batch = images).  Round 6: any kernel of up to 49 taps (31 on the 256-row tile, whose LDS
// holds the smaller tap table), any strides, any zero padding, any output width -- the reference's im2col is generic in all of
// them (conv2d_im2col.nim:42-88) -- and any Cin: a filter matrix whose rows are not whole 16-byte pieces (K % 4 != 0, or a strided
// view) is packed once into a zero-padded dense copy, which changes no bit (0 * 0 added to a chain).  hipErrorNotSupported: not
// this kernel's class.
hipError_t launch_conv_f32_asm(const GemmArgs<float> &a_in, bool laser_order, hipStream_t s) {
  GemmArgs<float> a = a_in;
  ConvClass cc;
  if (const hipError_t ce = conv_asm_classify(a, laser_order, cc); ce != hipSuccess) return ce;
  int pick = cc.pick;
  const KernelInfo &ki = kKernels[pick];
  const int tiles_m = cc.tiles_m, tiles_n = cc.tiles_n;
  const int64_t kH = cc.kH, kW = cc.kW, sH = cc.sH, sW = cc.sW, taps = cc.taps, oW = cc.oW, npix = cc.npix, Cin = cc.Cin, Kp = cc.Kp, tiles = cc.tiles;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  DeviceModule *m = nullptr;
  e = get_module(dev, &m);
  if (e != hipSuccess) return e;
  // the filter matrix as dense rows of Kp elements: as given when it already is (the usual case: K % 4 == 0, contiguous), else one
  // small packing pass into stream-ordered scratch (C_out x K floats: the first layer of a network has C_in = 3)
  float *packed = nullptr;
  if (a.csA != 1 || a.rsA != a.K || Kp != a.K) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return hipErrorNotSupported;
    if (a.rsA < 0 || a.csA < 0) return hipErrorNotSupported;
    e = scratch_alloc_async((void **)&packed, (size_t)a.M * Kp * 4, s);
    if (e != hipSuccess) return e;
    e = launch_pack_pad<float>(packed, a.M, Kp, a.A, a.M, a.K, a.rsA, a.csA, s, 0);
    if (e != hipSuccess) {
      (void)hipFreeAsync(packed, s);
      return e == hipErrorInvalidValue ? hipErrorNotSupported : e;
    }
  }
  // (one image's tiles are few: plain order of the tile ids -- tile rows fastest, no XCD remap -- keeps an image's pixels together
  // in an XCD's L2)
  const int group_m = 0;
  KernArgs ka;
  zero_conv_fields(ka);
  ka.A = packed ? packed : a.A;
  ka.B = a.B;
  ka.C = a.C;
  ka.unused_ = (const uint32_t *)(uintptr_t)magic_u32((uint64_t)taps);   // f32_kernel.py conv_load_ops: k / taps (KA_TAB's low word)
  ka.lda = (uint32_t)Kp;
  ka.ldb = 0;
  ka.ldc = (uint32_t)npix;
  ka.M = (uint32_t)a.M;
  ka.N = (uint32_t)a.N;
  ka.K = (uint32_t)Kp;
  ka.alpha = 1.0f;
  ka.beta = 0.0f;
  ka.H = (uint32_t)a.cH;
  ka.W = (uint32_t)a.cW;
  ka.oW = (uint32_t)oW;
  ka.pH = (uint32_t)a.cpH;
  ka.pW = (uint32_t)a.cpW;
  ka.Cin = (uint32_t)Cin;
  ka.Npix = (uint32_t)npix;
  ka.magic_oW = magic_u32((uint64_t)oW);   // floor(p / oW) = mulhi(p, magic) for p * oW < 2^32; 0 for oW == 1 (the kernel then takes p)
  ka.shift_oW = (uint32_t)(kH | kW << 8 | sH << 16 | sW << 24);     // geometry word (f32_kernel.py conv_setup)
  ka.pad_ = (uint32_t)(taps | ((1024 + kW - 1) / kW) << 16);         // taps | ceil(1024 / kW) << 16: kh = (r * that) >> 10
  ka.bsB_bytes = (uint64_t)a.bsB * 4;
  ka.bsC_bytes = (uint64_t)a.bsC * 4;
  ka.bias = a.bias;
  ka.rsBias = a.bias ? (uint32_t)a.rsBias : 0;
  ka.csBias = a.bias ? (uint32_t)a.csBias : 0;
  ka.act = (uint32_t)a.act;
  Plan plain;
  plain.G = tiles;
  // more units (images x tiles) than workgroup slots: the unit-walking form of the kernel -- every slot's workgroup goes through its
  // share of the units with pipelined transitions (option conv_walk: 1 = where it applies, 0 = never, 2 = whenever there are two units)
  const int64_t units = tiles * a.batch, slots = (int64_t)m->cus * ki.occ;
  const bool can_walk = a.bias == nullptr && a.act == 0 && Kp % 32 == 0 && Kp >= 96 && (double)units * (double)tiles < 4.0e9;
  int64_t walk_G = 0;
  if (can_walk && ((g_conv_walk == 1 && units > slots) || (g_conv_walk >= 2 && units >= 2))) {
    walk_G = std::min(slots, units);
    if (g_conv_walk >= 3) walk_G = std::min<int64_t>(g_conv_walk, units - 1);       // (tests: a forced workgroup count)
    pick = (pick == 10 || pick == 11 ? 66 + (pick - 10) : 68 + (pick - 21));
  }
  e = launch_planned(m, pick, plain, ka, tiles_m, tiles_n, group_m, a.batch, 0, s, walk_G);
  if (packed) {
    const hipError_t e2 = hipFreeAsync(packed, s);
    if (e == hipSuccess) e = e2;
  }
  if (e == hipSuccess) g_last_f32_asm = 1 + pick;
  return e;
}

}  // namespace laser_hip
