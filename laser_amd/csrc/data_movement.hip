// laser_amd/csrc/data_movement.hip -- the HBM-bound kernels around the GEMM: physical transposes
// (laser/primitives/swapaxes.nim:16-112), im2col (benchmarks/convolution/conv2d_im2col.nim:42-88)
// and the strided->dense panel copy used by the pre-pack API (gemm_prepacked.nim:87-218).
// No arithmetic: the only goals are fully coalesced 16-B-per-lane HBM traffic on both sides and
// enough workgroups (>> 256) to cover all CUs.
#include "common.h"

namespace laser_hip {

// ---- batched 2-D transpose: dst[n][j][i] = src[n][i][j] -------------------------------------------
// 64x64 tile through LDS ([64][65]: +1 pad makes the column reads conflict-free for 4-B elements),
// reads coalesced along NC, writes coalesced along NR (the reference writes contiguously and reads
// strided, swapaxes.nim:34-39; with LDS in between both sides are contiguous here).
template <typename T>
__global__ void __launch_bounds__(256) transpose_batched_kernel(T *__restrict__ dst, const T *__restrict__ src,
                                                                int64_t NR, int64_t NC, int64_t tiles_c,
                                                                int64_t tiles_r) {
  __shared__ T tile[64][65];
  const int64_t bid = blockIdx.x;
  const int64_t tc = bid % tiles_c, tr = (bid / tiles_c) % tiles_r, n = bid / (tiles_c * tiles_r);
  const int tx = threadIdx.x % 64, ty = threadIdx.x / 64;
  const T *s = src + n * NR * NC;
  T *d = dst + n * NR * NC;
  const int64_t r0 = tr * 64, c0 = tc * 64;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
    if (r < NR && c < NC) tile[ty + 4 * i][tx] = s[r * NC + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
    if (r < NR && c < NC) d[c * NR + r] = tile[tx][ty + 4 * i];
  }
}

hipError_t launch_transpose_batched(void *dst, const void *src, int64_t N, int64_t NR, int64_t NC,
                                    int elem_size, hipStream_t s) {
  if (N <= 0 || NR <= 0 || NC <= 0) return hipSuccess;
  const int64_t tiles_r = (NR + 63) / 64, tiles_c = (NC + 63) / 64;
  const int64_t blocks = N * tiles_r * tiles_c;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (elem_size == 4)
    hipLaunchKernelGGL(transpose_batched_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                       (uint32_t *)dst, (const uint32_t *)src, NR, NC, tiles_c, tiles_r);
  else if (elem_size == 8)
    hipLaunchKernelGGL(transpose_batched_kernel<uint64_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                       (uint64_t *)dst, (const uint64_t *)src, NR, NC, tiles_c, tiles_r);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---- im2col: [batch][C][H][W] -> [batch][C*kH*kW][oH*oW] -------------------------------------------
// One thread per workspace element; consecutive lanes run along ow (unit stride in both the
// workspace and, for stride 1, the input).  Same index arithmetic as conv2d_im2col.nim:62-87:
// row = -pH + krow + oh*sH, col = -pW + kcol + ow*sW, zero outside the image.
__global__ void __launch_bounds__(256) im2col_f32_kernel(float *__restrict__ ws, const float *__restrict__ in,
                                                         int64_t total, int C, int H, int W, int kH, int kW,
                                                         int oH, int oW, int pH, int pW, int sH, int sW) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int ow = (int)(r % oW); r /= oW;
    const int oh = (int)(r % oH); r /= oH;
    const int kcol = (int)(r % kW); r /= kW;
    const int krow = (int)(r % kH); r /= kH;
    const int c = (int)(r % C);
    const int64_t n = r / C;
    const int row = -pH + krow + oh * sH, col = -pW + kcol + ow * sW;
    float v = 0.0f;
    if (row >= 0 && row < H && col >= 0 && col < W) v = in[((n * C + c) * H + row) * (int64_t)W + col];
    ws[e] = v;
  }
}

hipError_t launch_im2col_f32(float *ws, int64_t oH, int64_t oW, const float *in, int64_t batch, int64_t C,
                             int64_t H, int64_t W, int64_t kH, int64_t kW, int64_t pH, int64_t pW,
                             int64_t sH, int64_t sW, hipStream_t s) {
  const int64_t total = batch * C * kH * kW * oH * oW;
  if (total <= 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  hipLaunchKernelGGL(im2col_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, in, total, (int)C,
                     (int)H, (int)W, (int)kH, (int)kW, (int)oH, (int)oW, (int)pH, (int)pW, (int)sH, (int)sW);
  return hipGetLastError();
}

// ---- strided -> dense zero-padded panel image (pre-pack) -------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) pack_pad_kernel(T *__restrict__ dst, int64_t Rpad, int64_t Cpad,
                                                       const T *__restrict__ src, int64_t R, int64_t Cc,
                                                       int64_t rs, int64_t cs) {
  const int64_t total = Rpad * Cpad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / Cpad, c = e % Cpad;
    dst[e] = (r < R && c < Cc) ? src[r * rs + c * cs] : (T)0;
  }
}

template <typename T>
hipError_t launch_pack_pad(T *dst, int64_t Rpad, int64_t Cpad, const T *src, int64_t R, int64_t Cc,
                           int64_t rs, int64_t cs, hipStream_t s) {
  const int64_t total = Rpad * Cpad;
  if (total <= 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pack_pad_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, dst, Rpad, Cpad, src, R,
                     Cc, rs, cs);
  return hipGetLastError();
}
template hipError_t launch_pack_pad<float>(float *, int64_t, int64_t, const float *, int64_t, int64_t, int64_t, int64_t, hipStream_t);
template hipError_t launch_pack_pad<double>(double *, int64_t, int64_t, const double *, int64_t, int64_t, int64_t, int64_t, hipStream_t);
template hipError_t launch_pack_pad<int32_t>(int32_t *, int64_t, int64_t, const int32_t *, int64_t, int64_t, int64_t, int64_t, hipStream_t);
template hipError_t launch_pack_pad<int64_t>(int64_t *, int64_t, int64_t, const int64_t *, int64_t, int64_t, int64_t, int64_t, hipStream_t);

}  // namespace laser_hip
