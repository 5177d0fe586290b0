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
 *     for K > 512).  f64 follows the same rule with kc = 256.
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
int laser_hip_device_count(void);
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
/* Convolution strategy: 1 (default) = implicit GEMM, im2col's index arithmetic fused into the GEMM's
 * B-tile loader (no workspace traffic); 0 = explicit im2col into the workspace + batched GEMM, the
 * reference's literal structure (conv2d_im2col.nim:126-166).  Results are bit-identical. */
int laser_hip_set_conv_implicit(int on);
/* 1 (default): the implicit conv reads its B operand from an LDS-resident input patch when that fits;
 * 0: always the per-element gather (A/B timing).  Results are bit-identical. */
int laser_hip_set_conv_patch(int on);
/* Host-pointer gemm_strided pipelines (A/B knob, default 1).  Bit 0: large calls on row-major-like operands with pinned B
 * and C run row panels of A x column panels of B, uploaded so the computable region grows as a square, every finished
 * strip of C copied back at once (the first kernel starts after one panel of each operand instead of after all of B);
 * clear = row panels only.  Bit 1: set = the small zero-copy path synchronises its stream instead of polling the
 * kernel's completion flags in mapped host memory.  Bit-identical. */
int laser_hip_set_host_pipeline(int mode);
/* 1 (default): the tail launch of a laser-order implicit conv (the output pixels past the last whole round of large
 * tiles) runs Laser's kc slices (gemm.nim:150-158) as parallel workgroup sets + an ordered combine; 0: one workgroup per
 * tail tile over all of K.  Results are bit-identical. */
int laser_hip_set_conv_kslice(int on);
/* 1 (default): float32/float64 problems with M <= 8 or N <= 8 (matrix-vector products) run a streaming kernel,
 * same arithmetic; 0: always the tiled kernels (A/B timing) */
int laser_hip_set_skinny(int on);
/* 1 (default): small float32 / float64 problems -- at most 256 blocks of 32x32 (f64: 16x16) outputs, or a batch of
 * matrices up to 64x64 -- run the small-matrix kernel (one wave per block of C, operands loaded straight into the
 * matrix-instruction registers, no LDS staging; the reference plans such a path: README.md:257-263): device-resident
 * operands for K <= 128 (BASELINE's 128^3), host-pointer calls for K <= 1024 and <= 1 MiB of operands, where the
 * kernel reads A / B from and writes C to a pinned staging buffer mapped into the device (one PCIe round trip
 * instead of three blocking copies).  Same arithmetic, same bits; 0: always the tiled kernels (A/B timing) */
int laser_hip_set_small_path(int on);
/* 1 (default): float problems with few output tiles and K >= 4 kc compute Laser's kc slices as one batched launch and
 * fold them with an ordered combine pass (same arithmetic, same order); 0: always the sequential K loop */
int laser_hip_set_slice_parallel(int on);
/* 1 (default): a float32 problem whose last round of workgroup tiles would be badly filled is cut along N into a
 * main launch (whole rounds of the large tile) and a tail launch (small tiles); tiles are independent and every
 * configuration computes identical bits, so results do not change.  0: always one launch; 2: the tail on a
 * library-owned side stream beside the main launch (event fork / join; measured slower, kept for A/B timing) */
int laser_hip_set_split_tail(int on);
/* diagnostics: the column where the last float GEMM / conv launch was cut (0: it ran as one launch) */
int64_t laser_hip_last_split(void);
/* diagnostics: index of the f32 tile configuration the last GEMM / conv launch used (-1: none yet; -2: the small-matrix kernel) */
int laser_hip_last_f32_config(void);
/* tuning knob for the transpose kernels' tile shape / streaming hints (0 = production form) */
int laser_hip_set_transpose_variant(int variant);
/* float32 gemm_strided with unit column strides, alpha == 1, beta == 0 and K a multiple of the K-tile: 1 (default) = the
 * hand-scheduled assembly kernels (one wave per SIMD, accumulators in AGPRs; laser_amd/asmgen/) when the tiles fill the
 * chip; 0 = always the compiler-scheduled kernels; 2 = whenever the problem is eligible (tests).  Bit-identical. */
int laser_hip_set_f32_asm(int on);
/* diagnostics: 0 = the last float32 GEMM launch was a compiler-scheduled kernel, 1 / 2 = the laser-order / fast assembly kernel */
int laser_hip_last_f32_asm(void);
/* int32 GEMM strategy: 1 (default) = signed 8-bit limb decomposition on the int8 matrix cores
 * (bit-exact mod 2^32); 0 = the VALU kernel.  Results are bit-identical. */
int laser_hip_set_i32_mfma(int on);
/* int64 GEMM strategy (the reference's int64 micro-kernel: gemm_ukernel_avx512.nim:58-74): 1 (default) = eight signed
 * 8-bit limbs, 36 limb products on the int8 matrix cores (bit-exact mod 2^64); 0 = the VALU kernel.  Bit-identical. */
int laser_hip_set_i64_mfma(int on);
/* float64 GEMM strategy: 1 (default) = v_mfma_f64_16x16x4_f64 (bitwise a k-ordered fma chain, so the
 * laser-order result is unchanged); 0 = the VALU kernel.  Results are bit-identical. */
int laser_hip_set_f64_mfma(int on);
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
 * (same doAssert as :125/:208 -> LASER_HIP_E_INVALID); the library uploads the operand ONCE into
 * a tile-padded device-resident panel image and writes a handle into the first 64 bytes of dst.
 * Device memory is released by laser_hip_gemm_prepack_release(dst) or laser_hip_finalize().
 * `_dev` variant: dst is a DEVICE buffer of `mem_required` bytes that receives the padded panel
 * image itself (no handle, nothing to release).  Unlike the reference (whose prepackA indexing is
 * only right for K <= kc, :186/:265) these are valid for every K. */
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
 * Pure data movement: one entry point per element size (b32: float32/int32, b64: float64/int64). */
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
/* One image [iC, iH, iW] -> workspace [iC*kH*kW, oH*oW]. */
int laser_hip_im2col_f32(float *pworkspace, int64_t oH, int64_t oW, const float *pinput, int64_t iC,
                         int64_t iH, int64_t iW, int64_t kH, int64_t kW, int64_t padH, int64_t padW,
                         int64_t strideH, int64_t strideW);
int laser_hip_im2col_f32_dev(float *d_workspace, int64_t oH, int64_t oW, const float *d_input,
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
 * alias a or b element for element.  alpha / beta are parameters of SCALE / FILL / AXPY / AXPBY (integers: exact up to
 * 2^53).  Element types f32, f64, i32, i64; the transcendental maps, RECIP and DIV are floating-point only
 * (LASER_HIP_E_INVALID otherwise).  HBM-bound, asynchronous on `stream`. */
#define LASER_HIP_MAP_COPY 0     /* a                     (= copy_strided) */
#define LASER_HIP_MAP_FILL 1     /* alpha                 (no operand read; `a` may be NULL) */
#define LASER_HIP_MAP_NEG 2
#define LASER_HIP_MAP_ABS 3
#define LASER_HIP_MAP_RELU 4     /* a > 0 ? a : 0 */
#define LASER_HIP_MAP_SCALE 5    /* alpha*a + beta        (two roundings) */
#define LASER_HIP_MAP_SQUARE 6
#define LASER_HIP_MAP_EXP 7
#define LASER_HIP_MAP_LOG 8
#define LASER_HIP_MAP_TANH 9
#define LASER_HIP_MAP_SIGMOID 10
#define LASER_HIP_MAP_SQRT 11
#define LASER_HIP_MAP_RECIP 12
#define LASER_HIP_MAP_ADD 32     /* binary from here on */
#define LASER_HIP_MAP_SUB 33
#define LASER_HIP_MAP_MUL 34
#define LASER_HIP_MAP_DIV 35
#define LASER_HIP_MAP_MAX 36
#define LASER_HIP_MAP_MIN 37
#define LASER_HIP_MAP_AXPY 38    /* alpha*a + b */
#define LASER_HIP_MAP_AXPBY 39   /* alpha*a + beta*b */
#define LASER_HIP_DECL_MAP(SFX, T)                                                                \
  int laser_hip_map_strided_unary_##SFX##_dev(int op, T *d_dst, const int64_t *dst_strides,        \
                                              const T *d_a, const int64_t *a_strides,              \
                                              const int64_t *shape, int rank, double alpha,        \
                                              double beta, void *stream);                          \
  int laser_hip_map_strided_binary_##SFX##_dev(int op, T *d_dst, const int64_t *dst_strides,       \
                                               const T *d_a, const int64_t *a_strides,             \
                                               const T *d_b, const int64_t *b_strides,             \
                                               const int64_t *shape, int rank, double alpha,       \
                                               double beta, void *stream);
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
/* Device-resident form (operands already in HBM; what the roofline is measured on).  Rows are dealt block-cyclically:
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
 *   flags         LASER_HIP_SHARD_PIN_TILE: 128x128 tiles for the local products (RCCL's kernels hold CUs meanwhile). */
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
