// tests/cpp/kat_host_mirror.cpp -- the reference's self-tests (gemm.nim:255-507) re-typed against the
// C++ host mirror include/laser.hpp, the way a compiled caller would use the drop-in.  Built and run
// by tests/test_gpu_cpp_mirror.py on the GPU box (g++, links liblaser_hip.so).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "laser.hpp"

template <typename T>
static int run() {
  int fails = 0;
  {  // gemm.nim:336-360  (M x K) * (K x N) with M < N
    const T a[2][3] = {{-2, -3, -1}, {3, 0, 4}};
    const T b[3][4] = {{1, 5, 2, -1}, {-3, 0, 3, 4}, {6, -2, 7, -4}};
    const T ab[2][4] = {{1, -8, -20, -6}, {27, 7, 34, -19}};
    T res[2][4];
    laser::gemm_strided<T>(2, 4, 3, T(1), &a[0][0], 3, 1, &b[0][0], 4, 1, T(0), &res[0][0], 4, 1);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) fails += (res[i][j] != ab[i][j]);
  }
  {  // gemm.nim:311-334
    const T a[2][3] = {{1, 2, 3}, {4, 5, 6}};
    const T b[3][2] = {{7, 8}, {9, 10}, {11, 12}};
    const T ab[2][2] = {{58, 64}, {139, 154}};
    T res[2][2];
    laser::gemm_strided<T>(2, 2, 3, T(1), &a[0][0], 3, 1, &b[0][0], 2, 1, T(0), &res[0][0], 2, 1);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) fails += (res[i][j] != ab[i][j]);
    // pack_and_test (gemm_prepacked.nim:314-351)
    void *pa = aligned_alloc(64, (size_t)((laser::gemm_prepackA_mem_required<T>(2, 2, 3) + 63) / 64 * 64));
    void *pb = aligned_alloc(64, (size_t)((laser::gemm_prepackB_mem_required<T>(2, 2, 3) + 63) / 64 * 64));
    laser::gemm_prepackA<T>(pa, 2, 2, 3, &a[0][0], 3, 1);
    laser::gemm_prepackB<T>(pb, 2, 2, 3, &b[0][0], 2, 1);
    T res2[2][2] = {{0, 0}, {0, 0}};
    laser::gemm_packed<T>(2, 2, 3, T(1), pa, pb, T(0), &res2[0][0], 2, 1);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) fails += (res2[i][j] != ab[i][j]);
    laser::gemm_prepack_release(pa);
    laser::gemm_prepack_release(pb);
    free(pa);
    free(pb);
  }
  return fails;
}

int main() {
  int fails = run<float>() + run<double>() + run<int32_t>() + run<int64_t>();
  // conv KAT 1 (conv2d_common.nim:147-186)
  const float in[16] = {1, 2, 0, 0, 5, 3, 0, 4, 0, 0, 0, 7, 9, 3, 0, 0};
  const float ker[9] = {1, 1, 1, 1, 1, 0, 1, 0, 0};
  const float target[16] = {1, 8, 5, 0, 8, 11, 5, 4, 8, 17, 10, 11, 9, 12, 10, 7};
  laser::TensorShape ishape{1, 1, 4, 4};
  laser::KernelShape kshape{1, 1, 3, 3};
  auto oshape = laser::conv2d_out_shape(ishape, kshape, {1, 1}, {1, 1});
  std::vector<float> out(16, 0.f), ws(laser::im2col_workspace_size(ishape, kshape, {1, 1}, {1, 1}));
  laser::conv2d_im2col(out.data(), oshape, in, ishape, ker, kshape, {1, 1}, {1, 1}, ws.data());
  for (int i = 0; i < 16; i++) fails += (out[i] != target[i]);
  // transposes
  const int32_t m[2][3] = {{1, 2, 3}, {4, 5, 6}};
  int32_t t[3][2];
  laser::transpose2D_copy(&t[0][0], &m[0][0], 2, 3);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) fails += (t[j][i] != m[i][j]);
  // device-resident Tensor chain (laser/tensor/*.nim twin): D = (A*B)^T * A, never leaving the GPU in between
  {
    const float a[2][3] = {{1, 2, 3}, {4, 5, 6}}, b[3][2] = {{7, 8}, {9, 10}, {11, 12}};
    auto A = laser::newTensor<float>({2, 3}), B = laser::newTensor<float>({3, 2});
    auto C = laser::newTensor<float>({2, 2}), D = laser::newTensor<float>({2, 3});
    laser::copyFromRaw(A, &a[0][0], 6);
    laser::copyFromRaw(B, &b[0][0], 6);
    laser::gemm(1.0f, A, B, 0.0f, C);                       // [[58, 64], [139, 154]]  (gemm.nim:311-334)
    laser::gemm(1.0f, C.transposed(), A, 0.0f, D);          // C^T * A through a strided view
    const float want[6] = {58 * 1 + 139 * 4, 58 * 2 + 139 * 5, 58 * 3 + 139 * 6, 64 * 1 + 154 * 4, 64 * 2 + 154 * 5, 64 * 3 + 154 * 6};
    auto d = D.to_host();
    for (int i = 0; i < 6; i++) fails += (d[i] != want[i]);
    auto ct = C.transposed().to_host();                     // deepCopy of a non-contiguous view
    fails += (ct[0] != 58) + (ct[1] != 139) + (ct[2] != 64) + (ct[3] != 154);
    fails += !C.is_C_contiguous() + C.transposed().is_C_contiguous() + (laser::newTensor<double>({4, 0, 2}).size() != 0);
    // forEach's device twin on a strided view: E = relu(C^T - 100), then E += C^T
    auto E = laser::newTensor<float>({2, 2});
    auto Ct = C.transposed();
    laser::forEachMap(LASER_HIP_MAP_SCALE, E, Ct, 1.0f, -100.0f);          // [[-42, 39], [-36, 54]]
    laser::forEachMap(LASER_HIP_MAP_RELU, E, E);                         // [[0, 39], [0, 54]]
    laser::forEachMap(LASER_HIP_MAP_ADD, E, E, Ct);                      // [[58, 178], [64, 208]]
    const float want_e[4] = {58, 178, 64, 208};
    auto e = E.to_host();
    for (int i = 0; i < 4; i++) fails += (e[i] != want_e[i]);
    laser::setZero(C);
    for (float v : C.to_host()) fails += (v != 0.f);
  }
  // error contract: misaligned pre-pack destination throws (reference: doAssert, gemm_prepacked.nim:125)
  bool threw = false;
  alignas(64) char buf[256];
  const float one = 1.f;
  try {
    laser::gemm_prepackB<float>(buf + 4, 1, 1, 1, &one, 1, 1);
  } catch (const laser::Error &) {
    threw = true;
  }
  fails += !threw;
  std::printf(fails ? "FAIL (%d)\n" : "SUCCESS\n", fails);
  return fails ? 1 : 0;
}
