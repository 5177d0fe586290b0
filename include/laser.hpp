// include/laser.hpp -- C++ host-side mirror of Laser's procs for the GEMM hot path, over the C-ABI
// of liblaser_hip.so (include/laser_hip.h).  Same names, argument order and meaning as the Nim
// originals (file:line per function); errors become exceptions where the reference doAsserts.
// Header-only; link with -llaser_hip.  No compute happens on the host.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "laser_hip.h"

namespace laser {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char *m) : std::runtime_error(std::string("laser_hip error ") + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
  if (rc != 0) throw Error(rc, laser_hip_last_error());
}

// shapes of benchmarks/convolution/conv2d_common.nim:6-13
struct TensorShape { int64_t n, c, h, w; };
struct KernelShape { int64_t c_out, c_in, kH, kW; };
struct Padding { int64_t h, w; };
struct Strides { int64_t h, w; };

#define LASER_DISPATCH(T, CALL_F32, CALL_F64, CALL_I32, CALL_I64)                         \
  if constexpr (std::is_same_v<T, float>) { CALL_F32; }                                   \
  else if constexpr (std::is_same_v<T, double>) { CALL_F64; }                             \
  else if constexpr (std::is_same_v<T, int32_t>) { CALL_I32; }                            \
  else { static_assert(std::is_same_v<T, int64_t>, "float, double, int32_t or int64_t"); CALL_I64; }

// gemm.nim:184-193
template <typename T>
void gemm_strided(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rowStrideA, int64_t colStrideA,
                  const T *B, int64_t rowStrideB, int64_t colStrideB, T beta, T *C, int64_t rowStrideC,
                  int64_t colStrideC) {
#define LASER_ARGS M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C, rowStrideC, colStrideC
  LASER_DISPATCH(T, check(laser_hip_gemm_strided_f32(LASER_ARGS)), check(laser_hip_gemm_strided_f64(LASER_ARGS)),
                 check(laser_hip_gemm_strided_i32(LASER_ARGS)), check(laser_hip_gemm_strided_i64(LASER_ARGS)))
#undef LASER_ARGS
}

// device-resident flavour (device pointers, hipStream_t as void*)
template <typename T>
void gemm_strided_dev(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *B,
                      int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC, void *stream = nullptr) {
#define LASER_ARGS M, N, K, alpha, A, rsA, csA, B, rsB, csB, beta, C, rsC, csC, stream
  LASER_DISPATCH(T, check(laser_hip_gemm_strided_f32_dev(LASER_ARGS)), check(laser_hip_gemm_strided_f64_dev(LASER_ARGS)),
                 check(laser_hip_gemm_strided_i32_dev(LASER_ARGS)), check(laser_hip_gemm_strided_i64_dev(LASER_ARGS)))
#undef LASER_ARGS
}

// Fused epilogue -- planned by the reference (README.md:238-242; TODO gemm.nim:196), not present in it:
//   C = act(alpha*A*B + beta*C + bias),  bias a strided M x N view (strides may be 0), float/double only.
enum class Activation : int { none = LASER_HIP_ACT_NONE, relu = LASER_HIP_ACT_RELU, tanh = LASER_HIP_ACT_TANH, sigmoid = LASER_HIP_ACT_SIGMOID };
template <typename T>
void gemm_strided_fused(int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rowStrideA, int64_t colStrideA,
                        const T *B, int64_t rowStrideB, int64_t colStrideB, T beta, T *C, int64_t rowStrideC,
                        int64_t colStrideC, const T *bias, int64_t rowStrideBias, int64_t colStrideBias,
                        Activation act = Activation::none) {
  static_assert(std::is_floating_point_v<T>, "the fused epilogue is float / double only");
  if constexpr (std::is_same_v<T, float>)
    check(laser_hip_gemm_strided_ex_f32(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C,
                                        rowStrideC, colStrideC, bias, rowStrideBias, colStrideBias, (int)act));
  else
    check(laser_hip_gemm_strided_ex_f64(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C,
                                        rowStrideC, colStrideC, bias, rowStrideBias, colStrideBias, (int)act));
}

// gemm_prepacked.nim:76-85, :157-167
template <typename T>
int64_t gemm_prepackB_mem_required(int64_t M, int64_t N, int64_t K) {
  LASER_DISPATCH(T, return laser_hip_gemm_prepackB_mem_required_f32(M, N, K), return laser_hip_gemm_prepackB_mem_required_f64(M, N, K),
                 return laser_hip_gemm_prepackB_mem_required_i32(M, N, K), return laser_hip_gemm_prepackB_mem_required_i64(M, N, K))
}
template <typename T>
int64_t gemm_prepackA_mem_required(int64_t M, int64_t N, int64_t K) {
  LASER_DISPATCH(T, return laser_hip_gemm_prepackA_mem_required_f32(M, N, K), return laser_hip_gemm_prepackA_mem_required_f64(M, N, K),
                 return laser_hip_gemm_prepackA_mem_required_i32(M, N, K), return laser_hip_gemm_prepackA_mem_required_i64(M, N, K))
}
// gemm_prepacked.nim:111-135, :193-218 (dst must be 64-byte aligned, like the reference's doAssert)
template <typename T>
void gemm_prepackB(void *dst_packedB, int64_t M, int64_t N, int64_t K, const T *src_B, int64_t rowStrideB, int64_t colStrideB) {
  LASER_DISPATCH(T, check(laser_hip_gemm_prepackB_f32(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)),
                 check(laser_hip_gemm_prepackB_f64(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)),
                 check(laser_hip_gemm_prepackB_i32(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)),
                 check(laser_hip_gemm_prepackB_i64(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)))
}
template <typename T>
void gemm_prepackA(void *dst_packedA, int64_t M, int64_t N, int64_t K, const T *src_A, int64_t rowStrideA, int64_t colStrideA) {
  LASER_DISPATCH(T, check(laser_hip_gemm_prepackA_f32(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)),
                 check(laser_hip_gemm_prepackA_f64(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)),
                 check(laser_hip_gemm_prepackA_i32(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)),
                 check(laser_hip_gemm_prepackA_i64(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)))
}
// gemm_prepacked.nim:275-292
template <typename T>
void gemm_packed(int64_t M, int64_t N, int64_t K, T alpha, const void *packedA, const void *packedB, T beta, T *C,
                 int64_t rowStrideC, int64_t colStrideC) {
  LASER_DISPATCH(T, check(laser_hip_gemm_packed_f32(M, N, K, alpha, packedA, packedB, beta, C, rowStrideC, colStrideC)),
                 check(laser_hip_gemm_packed_f64(M, N, K, alpha, packedA, packedB, beta, C, rowStrideC, colStrideC)),
                 check(laser_hip_gemm_packed_i32(M, N, K, alpha, packedA, packedB, beta, C, rowStrideC, colStrideC)),
                 check(laser_hip_gemm_packed_i64(M, N, K, alpha, packedA, packedB, beta, C, rowStrideC, colStrideC)))
}
inline void gemm_prepack_release(void *packed) { check(laser_hip_gemm_prepack_release(packed)); }

// swapaxes.nim:16-112
template <typename T>
void transpose2D_copy(T *dst, const T *src, int64_t NR, int64_t NC) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte elements");
  if constexpr (sizeof(T) == 4) check(laser_hip_transpose2d_copy_b32(dst, src, NR, NC));
  else check(laser_hip_transpose2d_copy_b64(dst, src, NR, NC));
}
template <typename T>
void transpose2D_batched(T *dst, const T *src, int64_t N, int64_t NR, int64_t NC) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte elements");
  if constexpr (sizeof(T) == 4) check(laser_hip_transpose2d_batched_b32(dst, src, N, NR, NC));
  else check(laser_hip_transpose2d_batched_b64(dst, src, N, NR, NC));
}
template <typename T>
void nchw2nhwc(T *dst_nhwc, const T *src_nchw, int64_t N, int64_t C, int64_t H, int64_t W) {
  transpose2D_batched(dst_nhwc, src_nchw, N, C, H * W);
}
template <typename T>
void nhwc2nchw(T *dst_nchw, const T *src_nhwc, int64_t N, int64_t C, int64_t H, int64_t W) {
  transpose2D_batched(dst_nchw, src_nhwc, N, H * W, C);
}

// conv2d_common.nim:15-45, conv2d_im2col.nim:10-166
inline TensorShape conv2d_out_shape(TensorShape i, KernelShape k, Padding p, Strides s) {
  TensorShape o{};
  check(laser_hip_conv2d_out_shape(i.n, i.c, i.h, i.w, k.c_out, k.c_in, k.kH, k.kW, p.h, p.w, s.h, s.w, &o.n, &o.c, &o.h, &o.w));
  return o;
}
inline int64_t im2col_workspace_size(TensorShape i, KernelShape k, Padding p, Strides s) {
  return laser_hip_im2col_workspace_size(i.n, i.c, i.h, i.w, k.c_out, k.c_in, k.kH, k.kW, p.h, p.w, s.h, s.w);
}
inline void im2col(float *pworkspace, TensorShape oshape, const float *pinput, TensorShape ishape, KernelShape kshape,
                   Padding padding, Strides strides) {
  check(laser_hip_im2col_f32(pworkspace, oshape.h, oshape.w, pinput, ishape.c, ishape.h, ishape.w, kshape.kH, kshape.kW,
                             padding.h, padding.w, strides.h, strides.w));
}
inline void conv2d_im2col(float *output, TensorShape oshape, const float *input, TensorShape ishape, const float *kernel,
                          KernelShape kshape, Padding padding, Strides strides, float *pworkspace) {
  if (oshape.c != kshape.c_out) throw Error(LASER_HIP_E_INVALID, "oshape.c != kshape.c_out");  // conv2d_im2col.nim:109
  check(laser_hip_conv2d_im2col_f32(output, input, ishape.n, ishape.c, ishape.h, ishape.w, kernel, kshape.c_out, kshape.c_in,
                                    kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w, pworkspace));
}

#undef LASER_DISPATCH
}  // namespace laser
