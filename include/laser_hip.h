/* include/laser_hip.h -- C-ABI of liblaser_hip.so, the MI355X (gfx950) implementation of Laser's
 * packed-panel GEMM hot path, its physical transposes and the im2col->GEMM convolution.
 *
 * This is the drop-in boundary: plain pointers, sizes and element strides; no C++/torch types.
 * Every entry point names the reference proc it replaces (file:line relative to the mratsim/laser
 * tree).  The Nim side keeps its own signatures and forwards here with
 *   {.dynlib: "liblaser_hip.so", importc: "laser_hip_...", cdecl.}
 * (see INTEGRATION.md and nim/laser_hip.nim).  Nim `int` == int64_t on amd64, `float32` == float.
 *
 * Conventions
 *   - Return value: 0 on success, non-zero on failure (LASER_HIP_E_*); the message is available
 *     from laser_hip_last_error() (thread-local).  The reference procs return void and abort via
 *     doAssert on precondition violations; the Nim shim turns a non-zero return into doAssert.
 *   - Strides are in ELEMENTS, element X[r,c] lives at ptr[r*rowStride + c*colStride]
 *     (gemm_utils.nim:36-60).  Any strides are accepted, including transposed and negative.
 *   - Host-pointer entry points (no suffix) are synchronous: they stage operands to the GPU, run,
 *     and copy C back before returning -- exactly Laser's blocking call semantics.  Caller keeps
 *     ownership of every pointer; the library owns only cached device scratch (freed by
 *     laser_hip_finalize).
 *   - `_dev` entry points take DEVICE pointers and a hipStream_t (as void*; NULL = default stream)
 *     and are asynchronous on that stream: this is the path measured against the roofline.
 *   - Semantics preserved from the reference: beta == 0 never reads C (NaN / uninitialised safe,
 *     gemm_ukernel_generic.nim:53-76); K == 0 leaves C untouched even if beta != 1 (gemm.nim:150);
 *     alpha, beta have the element type (integers too); overlapping A/B/C is undefined.
 *   - fp32 arithmetic order (LASER_HIP_F32_LASER_ORDER, the default): for every C[i,j] an
 *     ascending-k fused-multiply-add chain restarted from +0 every kc = 512 values of k, the slice
 *     sums added into beta*C in ascending order -- bit-identical to Laser on an FMA host
 *     (gemm_ukernel_generator.nim:245-248, gemm.nim:150-158, gemm_tiling.nim:309-310).
 *     LASER_HIP_F32_FAST keeps one chain across all of K (within 1e-5 relative, not bit-equal
 *     for K > 512; launch plans that split K -- few tiles x long K, the rows of a badly filled
 *     last round -- add kc-slice chains in order instead: the same bound).  f64 follows the same
 *     rule with kc = 256.
 */
#ifndef LASER_HIP_H
#define LASER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASER_HIP_OK 0
#define LASER_HIP_E_INVALID 1   /* bad argument (negative size, null pointer, misaligned pre-pack buffer ...) */
#define LASER_HIP_E_HIP 2       /* a HIP runtime call failed; see laser_hip_last_error() */
#define LASER_HIP_E_NODEVICE 3  /* no gfx950 device / library built without device code for this GPU */
#define LASER_HIP_E_HANDLE 4    /* pre-packed buffer handle is stale or corrupt */

/* ---- lifecycle ----------------------------------------------------------------------------- */
/* Lazy-initialised on first use; explicit init selects the device (-1 = current device). */
int laser_hip_init(int device);
int laser_hip_finalize(void);
const char *laser_hip_last_error(void);
const char *laser_hip_version(void);
/* Incremented whenever an exported prototype, an option name or an operation code changes incompatibly (round 3 changed the
 * alpha / beta types of map_strided and collapsed the setters into laser_hip_set_option: 2; round 4 only added names: still 2).
 * A caller built against another header should refuse to run. */
#define LASER_HIP_ABI_VERSION 2
int laser_hip_abi_version(void);
int laser_hip_device_count(void);
/* Diagnostics (touches no device): the kernel and launch plan the float32 launcher takes for a dense row-major M x N x K product on a
 * device of `cus` compute units -- the launch plans scale with the device's own CU count, so a partitioned MI355X (CPX 32 / QPX 64 /
 * DPX 128 CUs) keeps the hand-scheduled kernels.  out8[0] = 1 + kernel index (0: compiler-scheduled kernels), [1] = plan (0 one tile
 * per workgroup, 1 persistent with K-slice cuts, 2 strided whole tiles with pipelined transitions, 3 hybrid: 2 for the whole rounds + 1 for the remaining tiles), [2] = workgroups,
 * [3] = K slices per tile (hybrid: of the second launch), [4] = tiles, [5] / [6] = tile rows / columns, [7] = workgroup slots.  The GPU twin of reading gemm_tiling.nim:276-341's
 * `newTiles` for a shape: what partition will the library use. */
int laser_hip_plan_f32(int64_t M, int64_t N, int64_t K, int laser_order, int cus, int64_t *out8);
/* Name of the GPU architecture in use, e.g. "gfx950" (replaces the reference's cpuinfo ISA
 * dispatch, gemm.nim:228-247). */
const char *laser_hip_arch(void);

#define LASER_HIP_F32_LASER_ORDER 0
#define LASER_HIP_F32_FAST 1
int laser_hip_set_float_mode(int mode);
int laser_hip_get_float_mode(void);
/* Force one tile configuration of the f32 MFMA kernel (-1 = heuristic).  For tuning/benchmarks. */
int laser_hip_set_f32_config(int cfg);
int laser_hip_f32_config_count(void);
/* ---- options: one entry point for every tuning / A-B switch ------------------------------------------------------------
 * laser_hip_set_option(name, value); unknown names are LASER_HIP_E_INVALID.  In LASER_ORDER float mode (the default) every switch
 * leaves results bit-identical (it selects between implementations of the same arithmetic); in FAST mode the switches that change how
 * K is cut ("split_tail", "slice_parallel", "asm_plan" / "asm_slice" / "asm_wgs") change the rounding order of the sums, within the
 * 1e-5 mean relative error of that mode.  Defaults in brackets.
 *   "f32_asm"          [1] float32 gemm_strided with any element strides on A, B and C (MatrixView, gemm_utils.nim:36-60), any alpha /
 *                          beta, any K -- and 3x3 / stride-1 convolutions with any zero padding: the hand-scheduled assembly
 *                          kernels (accumulators in AGPRs; laser_amd/asmgen/) when the problem has at least ~100 tiles of 64x64;
 *                          0 = never (the compiler-scheduled kernels); 2 = whenever eligible, whatever the tile count (tests)
 *   "f64_asm"          [1] float64 twin of "f32_asm" (unit column strides on A and C, B row-major-like or passed transposed, any
 *                          alpha / beta, K even; laser_amd/asmgen/f64_kernel.py)
 *   "i32_asm"          [1] int32 / int64 limb GEMMs, any alpha / beta: the hand-scheduled kernels (laser_amd/asmgen/i8_kernel.py;
 *                          K > 8192 in chunks)
 *   "int_group_m"      [4] tile rows per raster group of those kernels' workgroup -> tile map (1 .. 64; a scheduling knob: same results)
 *   "f64_mfma" "i32_mfma" "i64_mfma"  [1] matrix-core kernels (f64 MFMA; int8-limb decomposition for the integers, the
 *                          reference's integer micro-kernels: gemm_ukernel_avx512.nim:40-74); 0 = the VALU kernels
 *   "conv_implicit"    [1] im2col fused into the GEMM's B loader; 0 = explicit im2col workspace + batched GEMM, the
 *                          reference's literal structure (conv2d_im2col.nim:126-166)
 *   "conv_patch"       [1] implicit conv reads B from an LDS-resident input patch when it fits; 0 = per-element gather
 *   "conv_direct"      [1] convolutions with <= 32 output channels and C_in*kH*kW <= 256 (the reference's conv bench shape,
 *                          conv2d_bench.nim:130-170): the direct HBM-streaming kernels; 0 = the implicit-GEMM kernels; 2 = without
 *                          the scalar-filter forms of 3x3 filters (A/B switch: the matrix-core / LDS-filter forms everywhere); 3 = the scalar-filter forms
 *                          without the two-channel-group split of small launches (A/B switch)
 *   "conv_tail"        [1] the pixel tail behind the hand-scheduled 3x3 conv main launch (npix % 128 pixels per image) as ONE launch
 *                          of the latency-built direct kernel (a wave per 32x32 block and kc slice, ordered fold in LDS); 0 = the
 *                          compiler-scheduled tail forms
 *   "conv_walk"        [1] assembly convolution main launch as unit walkers (a workgroup runs units (image, tile) g, g + G, ... with
 *                          pipelined transitions) where there are more units than workgroup slots; 0 never, 2 whenever there are two units,
 *                          >= 3: that many workgroups (tests).  Same bits either way.
 *   "conv_1x1_implicit" [0] probes: 1x1 / stride 1 / no padding convolutions through the implicit-GEMM kernels instead of the reference's
 *                          GEMM shortcut (conv2d_im2col.nim:121-153; measured 0.6 % ahead to 40 % behind it: profiles/r06/conv_1x1_probe_ae.jsonl)
 *   "conv_cut_always"  [0] tests / probes: cut every 3x3 convolution at its last whole 128-pixel tile, whatever the launch model says
 *   "conv_kslice"      [1] laser-order conv tail as parallel kc slices (gemm.nim:150-158) + ordered combine
 *   "host_pipeline_2d" [1] large host-pointer calls with pinned B and C: row panels x column panels; 0 = row panels only
 *   "zero_copy_poll"   [1] small host-pointer calls poll completion flags in mapped memory; 0 = synchronise the stream
 *   "skinny"           [1] M <= 8 or N <= 8: the streaming kernel        "small_path" [1] the one-wave-per-block kernel
 *   "split_tail"       [1] launch plans that cut a problem: main + tail launches of the compiler-scheduled kernels, peeled rows /
 *                          columns of ragged-by-a-few shapes, and the assembly kernels' persistent plan (below); 0 = one plain launch
 *   "slice_parallel"   [1] few tiles x long K: kc slices as one batched launch + ordered combine
 *   "slice_parallel_min" / "slice_parallel_tiles"  tuning overrides of that rule (0 = built-in)
 *   "asm_plan"         [0] launch plan of the assembly GEMM kernels: 0 = the launcher's model decides; 1 = one tile per workgroup;
 *                          2 = the persistent plan whenever legal: every workgroup slot of the chip gets an equal share of K-slice
 *                          units (laser-order: kc slices), a tile that straddles two workgroups is handed over in-kernel, in slice
 *                          order (gemm.nim:150-158) -- laser-order results are the same bits under every plan; 3 = the strided
 *                          whole-tile plan whenever legal (round 6): one persistent workgroup per slot walks tiles v, v + G, ...
 *                          WITHOUT leaving its K loop -- the last K-tile bodies of a tile fetch the next tile's first K-tiles, the next
 *                          tile's first body stores this tile's C (beta == 0, plain epilogue, K a multiple of the K-tile); what the
 *                          model takes by itself when there are more tiles than workgroup slots.  Same bits as plan 1, both modes.
 *                          4 = the hybrid plan whenever legal (round 6): two launches of one kernel -- the whole rounds of the tile
 *                          raster under plan 3, the remaining tiles (a fraction of a round, >= 8) under plan 2 -- what the model takes
 *                          in one-chain mode when the fraction is small; laser-order results are the same bits as under every plan.
 *                          A K-cut launch whose hand-over times out (a receiver polls ~2 s for its predecessor's running sum; never in
 *                          a correct run) is REPORTED: the error word travels back behind the launch and the next call on that stream
 *                          fails with LASER_HIP_E_HIP, laser_hip_last_error() naming the stream -- never a silently wrong C
 *                          ("asm_test_giveup" = 1 makes every receiver give up at once: tests of that report)
 *   "asm_kernel" [-1] / "asm_wgs" [0] / "asm_slice" [0] / "asm_noseed" [0] / "asm_group_m" [0]  tuning / test overrides of that plan:
 *                          force an assembly kernel index, the number of workgroups, the K-tiles per slice of a one-chain cut, the
 *                          two-run receive path, the tile rows per raster group (which tiles share an XCD's L2)
 *   "asm_tile"        [-1] pin a tile CLASS of the f32 assembly GEMM kernels (accumulation mode / transposed-B variant still follow the
 *                          call): 0 = 256x128 (laser-order) / 256x256, 1 = 256x128 one chain, 2 = 128x128x16 (two workgroups per CU:
 *                          degrades gracefully when another library's kernels -- RCCL's -- hold some CUs), 3 = 128x128x32, 4 = 64x64,
 *                          5 = 96x96, 6 = 160x96, 7 = 128x96, 8 = 192x96, 9 = 160x160 on 16x16 blocks (v_mfma_f32_16x16x4_f32: K % 4 == 0,
 *                          no fused prologue);
 *                          one tile per workgroup, no minimum tile count; -1 = the launcher's model decides.
 *   "thread_asm_tile" [-2] the same pin for the launches made BY THE CALLING THREAD only (-2 = none: "asm_tile" applies; -1 = the
 *                          model decides whatever "asm_tile" says).  The per-GPU processes of laser_amd/distributed.py set 2 around
 *                          their local products on the thread that launches them -- other threads' GEMMs keep their own choice;
 *                          LASER_HIP_SHARD_PIN_TILE is the same pin per call and per worker thread inside the single-process
 *                          sharded entry points
 *   "im2col_band"      [0] output pixels per workgroup band of the explicit im2col kernel (0 = 256 sixteen-byte vectors; tuning sweeps)
 * laser_hip_get_option reads any of them back, plus the read-only diagnostics of the last launch:
 *   "last_f32_config"  tile configuration index (-1 none yet, -2 small-matrix kernel, -3 direct small-channel conv kernel)
 *   "last_f32_asm" / "last_f64_asm" / "last_i32_asm"  0 = compiler-scheduled kernel, else 1 + index of the assembly kernel (gemm_f32_asm.cpp)
 *   "last_asm_wgs" / "last_asm_slices"  workgroups and K slices per tile of the last assembly launch (slices 1 = tiles never cut)
 *   "last_asm_rem"     tiles the last assembly launch left to the K-cut launch of a hybrid plan (0: one launch)
 *   "last_asm_group_m"  its raster group height in tile rows (which tiles share an XCD's L2), + 65536 when the workgroup ids were
 *                      chunked per XCD
 *   "asm_fixup_timeouts"  streams of the current device on which a workgroup of a cut launch gave up waiting for a hand-over (0 in a
 *                      correct run; reading it synchronises the device, and a stream reported here has its hand-over flags reset)
 *   "last_conv_tail"   how the last convolution's pixel tail ran: 0 no tail, 1 the direct tail kernel, 2 kc slices + combine, 3 one
 *                      compiler-kernel launch
 *   "last_split"       column where the last compiler-scheduled float GEMM / conv launch was cut into main + tail (0 = one launch) */
int laser_hip_set_option(const char *name, int value);
int laser_hip_get_option(const char *name, int64_t *value);
const char *laser_hip_f32_config_name(int cfg);

/* ---- gemm_strided -- laser/primitives/matrix_multiplication/gemm.nim:184-193 ------------------
 * proc gemm_strided*[T: SomeNumber](M, N, K: int, alpha: T, A: ptr T, rowStrideA, colStrideA: int,
 *        B: ptr T, rowStrideB, colStrideB: int, beta: T, C: ptr T, rowStrideC, colStrideC: int) */
#define LASER_HIP_DECL_GEMM(SFX, T)                                                               \
  int laser_hip_gemm_strided_##SFX(int64_t M, int64_t N, int64_t K, T alpha, const T *A,          \
                                   int64_t rowStrideA, int64_t colStrideA, const T *B,            \
                                   int64_t rowStrideB, int64_t colStrideB, T beta, T *C,          \
                                   int64_t rowStrideC, int64_t colStrideC);                       \
  int laser_hip_gemm_strided_##SFX##_dev(int64_t M, int64_t N, int64_t K, T alpha, const T *dA,   \
                                         int64_t rowStrideA, int64_t colStrideA, const T *dB,     \
                                         int64_t rowStrideB, int64_t colStrideB, T beta, T *dC,   \
                                         int64_t rowStrideC, int64_t colStrideC, void *stream);   \
  /* `batch` independent problems; operand b starts at ptr + b*batchStrideX (elements; 0 shares). \
   * Used by the convolution (one GEMM per image, conv2d_im2col.nim:126-166). */                  \
  int laser_hip_gemm_strided_batched_##SFX##_dev(                                                 \
      int64_t batch, int64_t M, int64_t N, int64_t K, T alpha, const T *dA, int64_t rowStrideA,   \
      int64_t colStrideA, int64_t batchStrideA, const T *dB, int64_t rowStrideB,                  \
      int64_t colStrideB, int64_t batchStrideB, T beta, T *dC, int64_t rowStrideC,                \
      int64_t colStrideC, int64_t batchStrideC, void *stream);
LASER_HIP_DECL_GEMM(f32, float)
LASER_HIP_DECL_GEMM(f64, double)
LASER_HIP_DECL_GEMM(i32, int32_t)
LASER_HIP_DECL_GEMM(i64, int64_t)
#undef LASER_HIP_DECL_GEMM

/* ---- pre-packed GEMM -- gemm_prepacked.nim:63-292 ----------------------------------------------
 * gemm_prepackB_mem_required*(T, M, N, K): int            :76-85
 * gemm_prepackB*[T](dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)   :111-135
 * gemm_prepackA_mem_required*, gemm_prepackA*             :157-218
 * gemm_packed*[T](M, N, K, alpha, packedA, packedB, beta, C, rowStrideC, colStrideC)  :275-292
 *
 * As in the reference the packed buffers are opaque, machine dependent and "unsafe to store or
 * serialize" (:120-123).  Host variant: the caller allocates `mem_required` bytes, 64-B aligned
 * (same doAssert as :125/:208 -> LASER_HIP_E_INVALID), and the buffer is SELF-CONTAINED like the
 * reference's (:111-135, Design.md:5-7): a 64-byte header followed by the tile-padded panel image.
 * It may be copied with memcpy and freed at will.  gemm_packed multiplies from a device-resident
 * copy of the image that the library caches (made by the prepack call; re-made from the caller's
 * buffer when it is missing; bounded -- least recently used first -- so freeing buffers without
 * telling the library cannot leak HBM).  laser_hip_gemm_prepack_release(dst) drops the cached copy
 * early and invalidates the buffer (optional); laser_hip_finalize() drops them all.
 * `_dev` variant: dst is a DEVICE buffer of `mem_required` bytes that receives the padded panel
 * image itself at offset 0 (no header, nothing cached).  Unlike the reference (whose prepackA
 * indexing is only right for K <= kc, :186/:265) these are valid for every K. */
#define LASER_HIP_DECL_PACK(SFX, T)                                                               \
  int64_t laser_hip_gemm_prepackA_mem_required_##SFX(int64_t M, int64_t N, int64_t K);            \
  int64_t laser_hip_gemm_prepackB_mem_required_##SFX(int64_t M, int64_t N, int64_t K);            \
  int laser_hip_gemm_prepackA_##SFX(void *dst_packedA, int64_t M, int64_t N, int64_t K,           \
                                    const T *src_A, int64_t rowStrideA, int64_t colStrideA);      \
  int laser_hip_gemm_prepackB_##SFX(void *dst_packedB, int64_t M, int64_t N, int64_t K,           \
                                    const T *src_B, int64_t rowStrideB, int64_t colStrideB);      \
  int laser_hip_gemm_packed_##SFX(int64_t M, int64_t N, int64_t K, T alpha, const void *packedA,  \
                                  const void *packedB, T beta, T *C, int64_t rowStrideC,          \
                                  int64_t colStrideC);                                            \
  int laser_hip_gemm_prepackA_##SFX##_dev(void *d_dst, int64_t M, int64_t N, int64_t K,           \
                                          const T *dA, int64_t rowStrideA, int64_t colStrideA,    \
                                          void *stream);                                          \
  int laser_hip_gemm_prepackB_##SFX##_dev(void *d_dst, int64_t M, int64_t N, int64_t K,           \
                                          const T *dB, int64_t rowStrideB, int64_t colStrideB,    \
                                          void *stream);                                          \
  int laser_hip_gemm_packed_##SFX##_dev(int64_t M, int64_t N, int64_t K, T alpha,                 \
                                        const void *d_packedA, const void *d_packedB, T beta,     \
                                        T *dC, int64_t rowStrideC, int64_t colStrideC,            \
                                        void *stream);
LASER_HIP_DECL_PACK(f32, float)
LASER_HIP_DECL_PACK(f64, double)
LASER_HIP_DECL_PACK(i32, int32_t)
LASER_HIP_DECL_PACK(i64, int64_t)
#undef LASER_HIP_DECL_PACK
int laser_hip_gemm_prepack_release(void *packed);

/* ---- physical transposes -- laser/primitives/swapaxes.nim:16-112 --------------------------------
 * transpose2D_copy*[T](dst, src, NR, NC)        :16-54   dst[j][i] = src[i][j]
 * transpose2D_batched*[T](dst, src, N, NR, NC)  :56-84
 * nchw2nhwc*[T](dst, src, N, C, H, W)           :86-98   = transpose2D_batched(N, C, H*W)
 * nhwc2nchw*[T](dst, src, N, C, H, W)           :100-112 = transpose2D_batched(N, H*W, C)
 * Pure data movement, generic in T like the reference: one entry point per element SIZE (b32: float32 / int32, b64: float64 /
 * int64, b16: float16 / int16, b8: int8 / uint8). */
#define LASER_HIP_DECL_TR(SFX)                                                                    \
  int laser_hip_transpose2d_copy_##SFX(void *dst, const void *src, int64_t NR, int64_t NC);       \
  int laser_hip_transpose2d_batched_##SFX(void *dst, const void *src, int64_t N, int64_t NR,      \
                                          int64_t NC);                                            \
  int laser_hip_nchw2nhwc_##SFX(void *dst, const void *src, int64_t N, int64_t C, int64_t H,      \
                                int64_t W);                                                       \
  int laser_hip_nhwc2nchw_##SFX(void *dst, const void *src, int64_t N, int64_t C, int64_t H,      \
                                int64_t W);                                                       \
  int laser_hip_transpose2d_batched_##SFX##_dev(void *d_dst, const void *d_src, int64_t N,        \
                                                int64_t NR, int64_t NC, void *stream);
LASER_HIP_DECL_TR(b32)
LASER_HIP_DECL_TR(b64)
LASER_HIP_DECL_TR(b16)
LASER_HIP_DECL_TR(b8)
#undef LASER_HIP_DECL_TR

/* ---- im2col + GEMM convolution -- benchmarks/convolution/conv2d_im2col.nim -----------------------
 * TensorShape = (n, c, h, w), KernelShape = (c_out, c_in, kH, kW), Padding = (h, w),
 * Strides = (h, w) (conv2d_common.nim:6-13) are passed as separate scalars.
 * conv2d_out_shape                           conv2d_common.nim:15-45
 * im2col_workspace_size*(ishape, kshape, padding, strides): int      conv2d_im2col.nim:10-20
 * im2col*[T](pworkspace, oshape, pinput, ishape, kshape, padding, strides)     :42-88
 * conv2d_im2col*(output, oshape, input, ishape, kernel, kshape, padding, strides, pworkspace) :90-166 */
int laser_hip_conv2d_out_shape(int64_t iN, int64_t iC, int64_t iH, int64_t iW, int64_t c_out,
                               int64_t c_in, int64_t kH, int64_t kW, int64_t padH, int64_t padW,
                               int64_t strideH, int64_t strideW, int64_t *oN, int64_t *oC,
                               int64_t *oH, int64_t *oW);
/* number of ELEMENTS, like the reference */
int64_t laser_hip_im2col_workspace_size(int64_t iN, int64_t iC, int64_t iH, int64_t iW,
                                        int64_t c_out, int64_t c_in, int64_t kH, int64_t kW,
                                        int64_t padH, int64_t padW, int64_t strideH,
                                        int64_t strideW);
/* One image [iC, iH, iW] -> workspace [iC*kH*kW, oH*oW].  im2col*[T] is generic in the reference (conv2d_im2col.nim:42-50):
 * _f32 and _f64 here (pure data movement; integer tensors of the same element size can be passed through a cast). */
int laser_hip_im2col_f32(float *pworkspace, int64_t oH, int64_t oW, const float *pinput, int64_t iC,
                         int64_t iH, int64_t iW, int64_t kH, int64_t kW, int64_t padH, int64_t padW,
                         int64_t strideH, int64_t strideW);
int laser_hip_im2col_f32_dev(float *d_workspace, int64_t oH, int64_t oW, const float *d_input,
                             int64_t batch, int64_t iC, int64_t iH, int64_t iW, int64_t kH,
                             int64_t kW, int64_t padH, int64_t padW, int64_t strideH,
                             int64_t strideW, void *stream);
int laser_hip_im2col_f64(double *pworkspace, int64_t oH, int64_t oW, const double *pinput, int64_t iC,
                         int64_t iH, int64_t iW, int64_t kH, int64_t kW, int64_t padH, int64_t padW,
                         int64_t strideH, int64_t strideW);
int laser_hip_im2col_f64_dev(double *d_workspace, int64_t oH, int64_t oW, const double *d_input,
                             int64_t batch, int64_t iC, int64_t iH, int64_t iW, int64_t kH,
                             int64_t kW, int64_t padH, int64_t padW, int64_t strideH,
                             int64_t strideW, void *stream);
/* output [iN, c_out, oH, oW] = conv(input [iN, iC, iH, iW], kernel [c_out, c_in, kH, kW]).
 * `pworkspace` (host variant) may be NULL: the scratch lives on the device; if non-NULL it must
 * hold im2col_workspace_size elements and receives the last image's im2col matrix, as in the
 * reference.  LASER_HIP_E_INVALID if c_in != iC (the reference asserts oshape.c == kshape.c_out,
 * :109, and conv2d_required_ops doAsserts C_in == kernel.c_in, conv2d_common.nim:66). */
int laser_hip_conv2d_im2col_f32(float *output, const float *input, int64_t iN, int64_t iC,
                                int64_t iH, int64_t iW, const float *kernel, int64_t c_out,
                                int64_t c_in, int64_t kH, int64_t kW, int64_t padH, int64_t padW,
                                int64_t strideH, int64_t strideW, float *pworkspace);
/* Device variant.  The default strategy (implicit GEMM) materialises nothing and ignores d_workspace.  On the
 * explicit path (kernels larger than 8x8, laser_hip_set_conv_implicit(0)) a non-NULL d_workspace is taken to hold
 * ONE image's im2col matrix -- im2col_workspace_size elements, the reference's contract ("can be reused between
 * batches", conv2d_im2col.nim:99) -- and the images are processed one by one through it; NULL = stream-ordered
 * library scratch (hipMallocAsync on `stream`), all images expanded in one pass and multiplied by one batched GEMM. */
int laser_hip_conv2d_im2col_f32_dev(float *d_output, const float *d_input, int64_t iN, int64_t iC,
                                    int64_t iH, int64_t iW, const float *d_kernel, int64_t c_out,
                                    int64_t c_in, int64_t kH, int64_t kW, int64_t padH,
                                    int64_t padW, int64_t strideH, int64_t strideW,
                                    float *d_workspace, void *stream);

/* ---- fused epilogue ------------------------------------------------
 * The reference plans it but does not have it: "fusing unary operations (like max/relu, tanh or
 * sigmoid) and binary operations (like adding a bias) at the end of the matrix multiplication
 * kernels" (README.md:238-242; TODOs gemm.nim:196, gemm_ukernel_generic.nim:78-79,128-129).
 *   C[i,j] = act( alpha*A*B + beta*C  +  bias[i*rowStrideBias + j*colStrideBias] )
 * The GEMM part is exactly gemm_strided (same accumulation order, same K == 0 rule: nothing is
 * touched); the bias is added with one more rounding after the LAST accumulation slice, then the
 * activation is applied -- once, on the accumulator, before the only store of C.  `bias` is a strided
 * M x N view whose strides may be 0: (1,0) = one value per row of C (a convolution's per-channel
 * bias), (0,1) = one per column (a dense layer's), (ldb,1) = a full residual matrix; NULL = none.
 * float32 / float64 only (the activations are floating-point functions). */
/* Fused PROLOGUE (the other half of the reference's fusion roadmap, README.md:243-244: "fuse operations before the matrix
 * multiplication kernel, during the prepacking ... for backward propagation"): OR these bits into `activation` and the product is
 * taken of relu(A) and / or relu(B), elementwise x > 0 ? x : 0, applied as the operand tile travels from HBM into the LDS panel
 * image (float32: the assembly kernels' `_pre` variants, in the staging registers; other paths: in the one packing pass the
 * operand gets anyway, or one made for it).  C = act(alpha * pre(A) * pre(B) + beta * C + bias). */
#define LASER_HIP_PRE_RELU_A 0x100
#define LASER_HIP_PRE_RELU_B 0x200
#define LASER_HIP_ACT_NONE 0
#define LASER_HIP_ACT_RELU 1     /* x > 0 ? x : 0 */
#define LASER_HIP_ACT_TANH 2
#define LASER_HIP_ACT_SIGMOID 3  /* 1 / (1 + exp(-x)) */
#define LASER_HIP_DECL_GEMM_EX(SFX, T)                                                            \
  int laser_hip_gemm_strided_ex_##SFX(int64_t M, int64_t N, int64_t K, T alpha, const T *A,       \
                                      int64_t rowStrideA, int64_t colStrideA, const T *B,         \
                                      int64_t rowStrideB, int64_t colStrideB, T beta, T *C,       \
                                      int64_t rowStrideC, int64_t colStrideC, const T *bias,      \
                                      int64_t rowStrideBias, int64_t colStrideBias,               \
                                      int activation);                                            \
  int laser_hip_gemm_strided_ex_##SFX##_dev(int64_t M, int64_t N, int64_t K, T alpha, const T *dA, \
                                            int64_t rowStrideA, int64_t colStrideA, const T *dB,  \
                                            int64_t rowStrideB, int64_t colStrideB, T beta, T *dC, \
                                            int64_t rowStrideC, int64_t colStrideC,               \
                                            const T *d_bias, int64_t rowStrideBias,               \
                                            int64_t colStrideBias, int activation, void *stream);
LASER_HIP_DECL_GEMM_EX(f32, float)
LASER_HIP_DECL_GEMM_EX(f64, double)
#undef LASER_HIP_DECL_GEMM_EX
/* conv2d_im2col with a per-output-channel bias ([c_out] or NULL) and an activation fused into the
 * implicit GEMM's store (signature = laser_hip_conv2d_im2col_f32[_dev] + bias, activation). */
int laser_hip_conv2d_im2col_ex_f32(float *output, const float *input, int64_t iN, int64_t iC,
                                   int64_t iH, int64_t iW, const float *kernel, int64_t c_out,
                                   int64_t c_in, int64_t kH, int64_t kW, int64_t padH, int64_t padW,
                                   int64_t strideH, int64_t strideW, float *pworkspace,
                                   const float *bias, int activation);
int laser_hip_conv2d_im2col_ex_f32_dev(float *d_output, const float *d_input, int64_t iN, int64_t iC,
                                       int64_t iH, int64_t iW, const float *d_kernel, int64_t c_out,
                                       int64_t c_in, int64_t kH, int64_t kW, int64_t padH,
                                       int64_t padW, int64_t strideH, int64_t strideW,
                                       float *d_workspace, const float *d_bias, int activation,
                                       void *stream);

/* ---- device tensor storage -----------------------------------------
 * The device twin of CpuStorage / allocCpuStorage and of the data-moving procs of
 * laser/tensor/initialization.nim, so that a Tensor[T]-shaped object (shape, strides, offset,
 * storage -- laser/tensor/datatypes.nim:18-30) can keep its buffer in HBM and chains of
 * gemm_strided / transposes / conv never cross PCIe.  The Tensor object itself stays on the host
 * language's side (nim/laser_hip.nim, include/laser.hpp, laser_amd/tensor.py); only raw buffers and
 * strides cross this ABI, as for every other entry point.
 *   storage_alloc    allocCpuStorage (allocator.nim:17-29): `bytes` of device memory, aligned to at
 *                    least LASER_MEM_ALIGN = 64 and zero-filled (allocShared0)
 *   storage_free     the storage finalizer (allocator.nim:11-15)
 *   storage_upload   copyFromRaw (initialization.nim:112-128): host buffer -> device storage
 *   storage_download the reverse (reading results back)
 *   storage_set_zero setZero (initialization.nim:130-154) on a contiguous extent
 *   copy_strided     `forEachStrided d in dst, s in src: d = s` (deepCopy / copyFrom of views,
 *                    initialization.nim:42-110): rank <= 6 (LASER_MAXRANK), element strides, same shape */
int laser_hip_storage_alloc(void **d_raw_buffer, int64_t bytes);
int laser_hip_storage_free(void *d_raw_buffer);
/* freed storages are cached per size for reuse (a device allocation costs ~100 us); this releases the cache */
int laser_hip_storage_trim(void);
int laser_hip_storage_upload(void *d_dst, const void *host_src, int64_t bytes);
int laser_hip_storage_download(void *host_dst, const void *d_src, int64_t bytes);
int laser_hip_storage_set_zero(void *d_buffer, int64_t bytes, void *stream);
/* Stream-ordered forms.  The plain storage_alloc completes its zero fill before returning and the plain upload /
 * download run on the NULL stream, which does NOT wait for non-blocking streams (PyTorch's side streams, any
 * hipStreamNonBlocking stream): a caller that computes on its own stream uses these so that the zero fill precedes
 * the first kernel on that stream, an upload follows the last reader and a download follows the producer.  Upload and
 * download are complete when the call returns. */
int laser_hip_storage_alloc_stream(void **d_raw_buffer, int64_t bytes, void *stream);
int laser_hip_storage_upload_stream(void *d_dst, const void *host_src, int64_t bytes, void *stream);
int laser_hip_storage_download_stream(void *host_dst, const void *d_src, int64_t bytes, void *stream);
int laser_hip_copy_strided_b32_dev(void *d_dst, const int64_t *dst_strides, const void *d_src,
                                   const int64_t *src_strides, const int64_t *shape, int rank,
                                   void *stream);
int laser_hip_copy_strided_b64_dev(void *d_dst, const int64_t *dst_strides, const void *d_src,
                                   const int64_t *src_strides, const int64_t *shape, int rank,
                                   void *stream);

/* ---- elementwise map over strided device views: the device twin of forEach ----------------------------------------
 * Laser's forEach / forEachStrided (laser/strided_iteration/foreach.nim:192-264) is a host macro over raw pointers
 * (`unsafe_raw_data`) with an odometer over shape / strides (foreach_common.nim:102-120).  A tensor whose storage lives
 * in HBM must never be handed to it (the host would dereference a device address: the Nim shim gives device storage a
 * distinct pointer type and no unsafe_raw_data for exactly that reason).  These entry points are what such tensors use
 * instead:      dst[idx] = f(a[idx])      /      dst[idx] = f(a[idx], b[idx])       for every index of `shape`
 * rank <= 6 (LASER_MAXRANK), every operand with its own ELEMENT strides (0 = broadcast along that dimension), dst may
 * alias a or b element for element.  alpha / beta (of the ELEMENT type, like gemm_strided's) are parameters of SCALE /
 * FILL / AXPY / AXPBY.  Element types f32, f64, i32, i64.  The ops are the arithmetic ones tensor initialisation needs
 * (laser/tensor/initialization.nim:42-154); SIMD exp / log and friends are out of scope.  HBM-bound,
 * asynchronous on `stream`. */
#define LASER_HIP_MAP_COPY 0     /* a                     (= copy_strided) */
#define LASER_HIP_MAP_FILL 1     /* alpha                 (no operand read; `a` may be NULL) */
#define LASER_HIP_MAP_NEG 2
#define LASER_HIP_MAP_ABS 3
#define LASER_HIP_MAP_RELU 4     /* a > 0 ? a : 0 */
#define LASER_HIP_MAP_SCALE 5    /* alpha*a + beta        (two roundings) */
#define LASER_HIP_MAP_SQUARE 6
#define LASER_HIP_MAP_ADD 32     /* binary from here on */
#define LASER_HIP_MAP_SUB 33
#define LASER_HIP_MAP_MUL 34
#define LASER_HIP_MAP_MAX 36
#define LASER_HIP_MAP_MIN 37
#define LASER_HIP_MAP_AXPY 38    /* alpha*a + b */
#define LASER_HIP_MAP_AXPBY 39   /* alpha*a + beta*b */
#define LASER_HIP_DECL_MAP(SFX, T)                                                                \
  int laser_hip_map_strided_unary_##SFX##_dev(int op, T *d_dst, const int64_t *dst_strides,        \
                                              const T *d_a, const int64_t *a_strides,              \
                                              const int64_t *shape, int rank, T alpha, T beta,     \
                                              void *stream);                                       \
  int laser_hip_map_strided_binary_##SFX##_dev(int op, T *d_dst, const int64_t *dst_strides,       \
                                               const T *d_a, const int64_t *a_strides,             \
                                               const T *d_b, const int64_t *b_strides,             \
                                               const int64_t *shape, int rank, T alpha, T beta,    \
                                               void *stream);
LASER_HIP_DECL_MAP(f32, float)
LASER_HIP_DECL_MAP(f64, double)
LASER_HIP_DECL_MAP(i32, int32_t)
LASER_HIP_DECL_MAP(i64, int64_t)
#undef LASER_HIP_DECL_MAP

/* ---- row-panel sharded gemm_strided over the GPUs of one node, ONE process ---------------------------------------
 * Laser partitions M across its OpenMP threads with no cross-thread reduction (gemm.nim:160-176: `omp for` over the
 * ic row blocks), so rows of C are independent units: each GPU owns row panels of A, all of B, and produces the
 * matching rows of C.  No K split => no reduction => every element is computed exactly as on one GPU (bit-identical,
 * whatever the device count).  `devices`: ndev HIP ordinals, or NULL for 0..ndev-1; ndev <= 0 = every visible GPU.
 * Both forms are synchronous (return when every device is done); one host thread per GPU drives it.
 *
 * Host pointers (the drop-in form -- same parameter list as gemm_strided after the device list): the rows are cut
 * into one contiguous range per GPU, each GPU runs the ordinary host-pointer pipeline on its own PCIe link and writes
 * its rows of C straight back to host memory (no inter-GPU traffic).  laser_hip_set_shard_devices(n) makes the PLAIN
 * laser_hip_gemm_strided_* entry points route large problems here (n = 1: off, the default; 0: every GPU), so an
 * unchanged Nim gemm_strided call uses the node. */
int laser_hip_set_shard_devices(int ndev);
int laser_hip_get_shard_devices(void);
/* Device-resident form (operands already in HBM; what the roofline is measured on).  The call has no stream parameter: it works on
 * the library's own streams (one set per device slot), so every operand must be COMPLETE in memory when it is made -- a caller that
 * fills A / B / C asynchronously on its own stream synchronises that stream first (the Python mirror does).  Rows are dealt block-cyclically:
 * panel (s, g) -- sub-panel s of device slot g -- is rows [(s*ndev + g)*R, +R) of C, R = rows_per_panel from
 * laser_hip_shard_plan (a multiple of 256 when M allows; panels_per_dev is reduced if steps would be empty).
 *   dA_panels[g]  device slot g's panels of A stacked in local order: row s*R + i = global row (s*ndev + g)*R + i
 *                 (element strides rowStrideA / colStrideA)
 *   dB[g]         B, replicated on every device (strides rowStrideB / colStrideB)
 *   dC[g]         the FULL row-major C on every device (colStride 1, rowStrideC >= N): device g computes its panels in
 *                 place and, with a gather mode, receives everybody else's -- the all-gather of C over xGMI -- sub-panel
 *                 s being sent while sub-panel s+1 multiplies.  beta != 0 reads the device's own copy of its rows.
 *   gather        LASER_HIP_GATHER_NONE  every device keeps only its own rows;
 *                 LASER_HIP_GATHER_PEER  the owner pushes finished rows to every peer with hipMemcpyPeerAsync on one
 *                                        copy stream per peer (all xGMI links at once, SDMA engines, no CUs);
 *                 LASER_HIP_GATHER_RCCL  ncclAllGather per slab (librccl.so loaded on first use; needs
 *                                        rowStrideC == N and dC[g] sized for padded_M rows).
 *   flags         LASER_HIP_SHARD_PIN_TILE: the hand-scheduled 128x128x16 assembly tile (option "asm_tile" = 2) for the local f32
 *                 products of this call (RCCL's kernels hold CUs meanwhile); other dtypes ignore it.
 * On an error return the operand buffers must stay alive until the devices are idle (work queued by the failed call may still
 * name them); the library drops the streams it queued that work on and makes new ones for the next call. */
#define LASER_HIP_GATHER_NONE 0
#define LASER_HIP_GATHER_PEER 1
#define LASER_HIP_GATHER_RCCL 2
#define LASER_HIP_SHARD_PIN_TILE 1
int laser_hip_shard_plan(int64_t M, int ndev, int panels_per_dev, int64_t *rows_per_panel,
                         int *panels_per_dev_used, int64_t *padded_M);
#define LASER_HIP_DECL_SHARDED(SFX, T)                                                            \
  int laser_hip_gemm_strided_##SFX##_sharded(int ndev, const int *devices, int64_t M, int64_t N,  \
                                             int64_t K, T alpha, const T *A, int64_t rowStrideA,  \
                                             int64_t colStrideA, const T *B, int64_t rowStrideB,  \
                                             int64_t colStrideB, T beta, T *C, int64_t rowStrideC,\
                                             int64_t colStrideC);                                 \
  int laser_hip_gemm_strided_##SFX##_sharded_dev(                                                 \
      int ndev, const int *devices, int64_t M, int64_t N, int64_t K, T alpha,                     \
      const T *const *dA_panels, int64_t rowStrideA, int64_t colStrideA, const T *const *dB,      \
      int64_t rowStrideB, int64_t colStrideB, T beta, T *const *dC, int64_t rowStrideC,           \
      int panels_per_dev, int gather, int flags);
LASER_HIP_DECL_SHARDED(f32, float)
LASER_HIP_DECL_SHARDED(f64, double)
LASER_HIP_DECL_SHARDED(i32, int32_t)
LASER_HIP_DECL_SHARDED(i64, int64_t)
#undef LASER_HIP_DECL_SHARDED

/* ---- pinned host memory (optional) ---------------------------------------------------------------------------------
 * The host-pointer entry points accept ANY host memory.  From pageable memory the runtime stages or pins every copy on
 * the fly; a tensor allocator that wants the full PCIe rate allocates its buffers with host_alloc (64-byte aligned, as
 * LASER_MEM_ALIGN asks) or registers existing ones -- Laser leaves buffer management to the caller (Design.md:5-7).
 * The same entry points then move the operands by direct asynchronous DMA; results are identical. */
int laser_hip_host_alloc(void **host_ptr, int64_t bytes);
int laser_hip_host_free(void *host_ptr);
int laser_hip_host_register(void *host_ptr, int64_t bytes);
int laser_hip_host_unregister(void *host_ptr);

/* ---- cblas-shaped GEMM -- benchmarks/third_party/blas.nim:12-23 ---------------------------------
 * The call conv2d_im2col makes (conv2d_im2col.nim:161-166); ORDER 101 rowMajor / 102 colMajor,
 * TRANS 111 noTranspose / 112 transpose / 113 conjTranspose.  Mapped onto gemm_strided strides. */
int laser_hip_cblas_sgemm(int order, int transA, int transB, int64_t M, int64_t N, int64_t K,
                          float alpha, const float *A, int64_t lda, const float *B, int64_t ldb,
                          float beta, float *C, int64_t ldc);
int laser_hip_cblas_dgemm(int order, int transA, int transB, int64_t M, int64_t N, int64_t K,
                          double alpha, const double *A, int64_t lda, const double *B, int64_t ldb,
                          double beta, double *C, int64_t ldc);

#ifdef __cplusplus
}
#endif
#endif /* LASER_HIP_H */
