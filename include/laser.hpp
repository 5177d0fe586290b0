// include/laser.hpp -- C++ host-side mirror of Laser's procs for the GEMM hot path, over the C-ABI
// of liblaser_hip.so (include/laser_hip.h).  Same names, argument order and meaning as the Nim
// originals (file:line per function); errors become exceptions where the reference doAsserts.
// Header-only; link with -llaser_hip.  No compute happens on the host.
#pragma once
#include <array>
#include <cstdint>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <vector>
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
  static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8, "1-, 2-, 4- or 8-byte elements");
  if constexpr (sizeof(T) == 4) check(laser_hip_transpose2d_copy_b32(dst, src, NR, NC));
  else if constexpr (sizeof(T) == 8) check(laser_hip_transpose2d_copy_b64(dst, src, NR, NC));
  else if constexpr (sizeof(T) == 2) check(laser_hip_transpose2d_copy_b16(dst, src, NR, NC));
  else check(laser_hip_transpose2d_copy_b8(dst, src, NR, NC));
}
template <typename T>
void transpose2D_batched(T *dst, const T *src, int64_t N, int64_t NR, int64_t NC) {
  static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8, "1-, 2-, 4- or 8-byte elements");
  if constexpr (sizeof(T) == 4) check(laser_hip_transpose2d_batched_b32(dst, src, N, NR, NC));
  else if constexpr (sizeof(T) == 8) check(laser_hip_transpose2d_batched_b64(dst, src, N, NR, NC));
  else if constexpr (sizeof(T) == 2) check(laser_hip_transpose2d_batched_b16(dst, src, N, NR, NC));
  else check(laser_hip_transpose2d_batched_b8(dst, src, N, NR, NC));
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
// (conv2d_im2col.nim:42-50: generic in T; pure data movement, so dispatched on the element size)
template <typename T>
void im2col(T *pworkspace, TensorShape oshape, const T *pinput, TensorShape ishape, KernelShape kshape, Padding padding,
            Strides strides) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte elements");
  if constexpr (sizeof(T) == 4)
    check(laser_hip_im2col_f32(reinterpret_cast<float *>(pworkspace), oshape.h, oshape.w, reinterpret_cast<const float *>(pinput),
                               ishape.c, ishape.h, ishape.w, kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w));
  else
    check(laser_hip_im2col_f64(reinterpret_cast<double *>(pworkspace), oshape.h, oshape.w, reinterpret_cast<const double *>(pinput),
                               ishape.c, ishape.h, ishape.w, kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w));
}
inline void conv2d_im2col(float *output, TensorShape oshape, const float *input, TensorShape ishape, const float *kernel,
                          KernelShape kshape, Padding padding, Strides strides, float *pworkspace) {
  if (oshape.c != kshape.c_out) throw Error(LASER_HIP_E_INVALID, "oshape.c != kshape.c_out");  // conv2d_im2col.nim:109
  check(laser_hip_conv2d_im2col_f32(output, input, ishape.n, ishape.c, ishape.h, ishape.w, kernel, kshape.c_out, kshape.c_in,
                                    kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w, pworkspace));
}

// ---- device-resident Tensor[T] -- laser/tensor/datatypes.nim:18-52, initialization.nim:42-202 ------
// Same fields as the reference's Tensor (shape, strides, offset, storage); the storage's raw_buffer is a
// DEVICE address, so gemm_strided_dev / transposes / conv chained on Tensors never cross PCIe.
constexpr int LASER_MAXRANK = 6;  // laser/dynamic_stack_arrays.nim:6

template <typename T>
struct HipStorage {  // CpuStorage's twin (datatypes.nim:24-30); freed by the destructor = the Nim finalizer
  T *raw_buffer = nullptr;
  void *memalloc = nullptr;
  bool memowner = false;
  explicit HipStorage(int64_t size) {  // allocCpuStorage (allocator.nim:17-29): aligned, zero-filled
    check(laser_hip_storage_alloc(&memalloc, size * (int64_t)sizeof(T)));
    raw_buffer = static_cast<T *>(memalloc);
    memowner = true;
  }
  HipStorage(const HipStorage &) = delete;
  HipStorage &operator=(const HipStorage &) = delete;
  ~HipStorage() {
    if (memowner && memalloc) (void)laser_hip_storage_free(memalloc);
  }
};

template <typename T>
struct Tensor {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float, double, int32_t or int64_t");
  std::vector<int64_t> shape, strides;
  int64_t offset = 0;
  std::shared_ptr<HipStorage<T>> storage;

  int rank() const { return (int)shape.size(); }
  int64_t size() const {
    int64_t n = 1;
    for (int64_t s : shape) n *= s;
    return n;
  }
  bool is_C_contiguous() const {  // datatypes.nim:38-47
    int64_t cur = 1;
    for (int i = rank() - 1; i >= 0; i--) {
      if (shape[i] != 1 && strides[i] != cur) return false;
      cur *= shape[i];
    }
    return true;
  }
  T *unsafe_raw_data() { return storage->raw_buffer + offset; }              // device pointer
  const T *unsafe_raw_data() const { return storage->raw_buffer + offset; }
  Tensor transposed() const {  // view: reversed axes
    Tensor t = *this;
    t.shape.assign(shape.rbegin(), shape.rend());
    t.strides.assign(strides.rbegin(), strides.rend());
    return t;
  }
  std::vector<T> to_host() const;
};

template <typename T>
Tensor<T> newTensor(std::initializer_list<int64_t> shape) {  // initialization.nim:156-166 (zero-initialised)
  if ((int)shape.size() > LASER_MAXRANK) throw Error(LASER_HIP_E_INVALID, "rank > LASER_MAXRANK");
  Tensor<T> t;
  t.shape.assign(shape.begin(), shape.end());
  t.strides.resize(t.shape.size());
  int64_t size = 1;
  for (int i = t.rank() - 1; i >= 0; i--) {
    t.strides[i] = size;
    size *= t.shape[i];
  }
  t.storage = std::make_shared<HipStorage<T>>(size);
  return t;
}
template <typename T>
void copy_strided_dev(Tensor<T> &dst, const Tensor<T> &src) {
  if (dst.shape != src.shape) throw Error(LASER_HIP_E_INVALID, "shapes differ");
  if constexpr (sizeof(T) == 4)
    check(laser_hip_copy_strided_b32_dev(dst.unsafe_raw_data(), dst.strides.data(), src.unsafe_raw_data(), src.strides.data(),
                                         dst.shape.data(), dst.rank(), nullptr));
  else
    check(laser_hip_copy_strided_b64_dev(dst.unsafe_raw_data(), dst.strides.data(), src.unsafe_raw_data(), src.strides.data(),
                                         dst.shape.data(), dst.rank(), nullptr));
}
template <typename T>
void copyFrom(Tensor<T> &dst, const Tensor<T> &src) { copy_strided_dev(dst, src); }  // initialization.nim:77-110
template <typename T>
void deepCopy(Tensor<T> &dst, const Tensor<T> &src) {  // initialization.nim:42-75: fresh row-major storage
  Tensor<T> fresh;
  fresh.shape = src.shape;
  fresh.strides.resize(src.shape.size());
  int64_t size = 1;
  for (int i = fresh.rank() - 1; i >= 0; i--) {
    fresh.strides[i] = size;
    size *= fresh.shape[i];
  }
  fresh.storage = std::make_shared<HipStorage<T>>(size);
  copy_strided_dev(fresh, src);
  dst = fresh;
}
template <typename T>
void copyFromRaw(Tensor<T> &dst, const T *buffer, int64_t len) {  // initialization.nim:112-128
  if (dst.size() != len) throw Error(LASER_HIP_E_INVALID, "Tensor size and buffer length should be the same");
  if (!dst.is_C_contiguous()) throw Error(LASER_HIP_E_INVALID, "copyFromRaw needs a contiguous destination");
  check(laser_hip_storage_upload(dst.unsafe_raw_data(), buffer, len * (int64_t)sizeof(T)));
}
template <typename T>
void setZero(Tensor<T> &t) {  // initialization.nim:130-154
  if (!t.is_C_contiguous()) throw Error(LASER_HIP_E_INVALID, "Input tensor is not contiguous.");
  check(laser_hip_storage_set_zero(t.unsafe_raw_data(), t.size() * (int64_t)sizeof(T), nullptr));
}
template <typename T>
std::vector<T> Tensor<T>::to_host() const {
  Tensor<T> c;
  const Tensor<T> *src = this;
  if (!is_C_contiguous()) {
    deepCopy(c, *this);
    src = &c;
  }
  std::vector<T> out((size_t)size());
  check(laser_hip_storage_download(out.data(), src->unsafe_raw_data(), size() * (int64_t)sizeof(T)));
  return out;
}
// forEach's device twin (laser/strided_iteration/foreach.nim:192-264): dst[idx] = f(a[idx]) / f(a[idx], b[idx]) over strided
// views of one shape (b may broadcast through zero strides set by the caller); op = LASER_HIP_MAP_* of laser_hip.h
template <typename T>
void forEachMap(int op, Tensor<T> &dst, const Tensor<T> &a, T alpha = T(1), T beta = T(0)) {
  if (dst.shape != a.shape) throw Error(LASER_HIP_E_INVALID, "shapes differ");
#define LASER_ARGS op, dst.unsafe_raw_data(), dst.strides.data(), a.unsafe_raw_data(), a.strides.data(), dst.shape.data(), dst.rank(), alpha, beta, nullptr
  LASER_DISPATCH(T, check(laser_hip_map_strided_unary_f32_dev(LASER_ARGS)), check(laser_hip_map_strided_unary_f64_dev(LASER_ARGS)),
                 check(laser_hip_map_strided_unary_i32_dev(LASER_ARGS)), check(laser_hip_map_strided_unary_i64_dev(LASER_ARGS)))
#undef LASER_ARGS
}
template <typename T>
void forEachMap(int op, Tensor<T> &dst, const Tensor<T> &a, const Tensor<T> &b, T alpha = T(1), T beta = T(1)) {
  if (dst.shape != a.shape || dst.shape != b.shape) throw Error(LASER_HIP_E_INVALID, "shapes differ");
#define LASER_ARGS op, dst.unsafe_raw_data(), dst.strides.data(), a.unsafe_raw_data(), a.strides.data(), b.unsafe_raw_data(), b.strides.data(), dst.shape.data(), dst.rank(), alpha, beta, nullptr
  LASER_DISPATCH(T, check(laser_hip_map_strided_binary_f32_dev(LASER_ARGS)), check(laser_hip_map_strided_binary_f64_dev(LASER_ARGS)),
                 check(laser_hip_map_strided_binary_i32_dev(LASER_ARGS)), check(laser_hip_map_strided_binary_i64_dev(LASER_ARGS)))
#undef LASER_ARGS
}
// every tuning / A-B switch by name, and the diagnostics of the last launch (include/laser_hip.h lists them)
inline void set_option(const char *name, int value) { check(laser_hip_set_option(name, value)); }
inline int64_t get_option(const char *name) {
  int64_t v = 0;
  check(laser_hip_get_option(name, &v));
  return v;
}
// the node behind an unchanged host-pointer gemm_strided call (0 = every visible GPU, 1 = off)
inline void set_shard_devices(int ndev) { check(laser_hip_set_shard_devices(ndev)); }

// One gemm_strided over every GPU with the operands resident in HBM and C all-gathered over xGMI
// (laser_hip_gemm_strided_*_sharded_dev): block-cyclic row panels, no K split -- bit-identical to one GPU.
struct ShardPlan {
  int64_t rows_per_panel = 0, padded_M = 0;
  int panels_per_dev = 0;
};
inline ShardPlan shard_plan(int64_t M, int ndev, int panels_per_dev = 4) {
  ShardPlan p;
  check(laser_hip_shard_plan(M, ndev, panels_per_dev, &p.rows_per_panel, &p.panels_per_dev, &p.padded_M));
  return p;
}
template <typename T>
void gemm_strided_sharded_dev(const std::vector<int> &devices, int64_t M, int64_t N, int64_t K, T alpha,
                              const std::vector<const T *> &dA_panels, int64_t rsA, int64_t csA, const std::vector<const T *> &dB,
                              int64_t rsB, int64_t csB, T beta, const std::vector<T *> &dC, int64_t rsC, int panels_per_dev = 4,
                              int gather = LASER_HIP_GATHER_PEER, int flags = 0) {
  if (dA_panels.size() != devices.size() || dB.size() != devices.size() || dC.size() != devices.size())
    throw Error(LASER_HIP_E_INVALID, "gemm_strided_sharded_dev: one pointer per device slot");
#define LASER_ARGS (int)devices.size(), devices.data(), M, N, K, alpha, dA_panels.data(), rsA, csA, dB.data(), rsB, csB, beta, dC.data(), rsC, panels_per_dev, gather, flags
  LASER_DISPATCH(T, check(laser_hip_gemm_strided_f32_sharded_dev(LASER_ARGS)), check(laser_hip_gemm_strided_f64_sharded_dev(LASER_ARGS)),
                 check(laser_hip_gemm_strided_i32_sharded_dev(LASER_ARGS)), check(laser_hip_gemm_strided_i64_sharded_dev(LASER_ARGS)))
#undef LASER_ARGS
}

// C (M x N) = alpha * A (M x K) * B (K x N) + beta * C on 2-D Tensors of any strides, all device-resident
template <typename T>
void gemm(T alpha, const Tensor<T> &A, const Tensor<T> &B, T beta, Tensor<T> &C) {
  if (A.rank() != 2 || B.rank() != 2 || C.rank() != 2 || A.shape[1] != B.shape[0] || C.shape[0] != A.shape[0] ||
      C.shape[1] != B.shape[1])
    throw Error(LASER_HIP_E_INVALID, "gemm: shapes do not agree");
  gemm_strided_dev<T>(A.shape[0], B.shape[1], A.shape[1], alpha, A.unsafe_raw_data(), A.strides[0], A.strides[1],
                      B.unsafe_raw_data(), B.strides[0], B.strides[1], beta, C.unsafe_raw_data(), C.strides[0], C.strides[1]);
}

#undef LASER_DISPATCH
}  // namespace laser
