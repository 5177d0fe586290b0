// laser_amd/csrc/map_strided.hip -- elementwise map over strided rank <= 6 device views: the device-side twin of
// Laser's forEach / forEachStrided (laser/strided_iteration/foreach.nim:192-264, foreach_common.nim:102-120).
//
// The host forEach is a macro that takes raw pointers (`unsafe_raw_data`) and walks them on the CPU with an odometer
// over shape / strides; a tensor whose storage lives in HBM must never reach it (the pointer is a device address).
// What tensor initialisation and the GEMM path's call sites need from it (laser/tensor/initialization.nim:42-154: copy,
// fill, scale, axpy; plus neg / abs / relu / square / add / sub / mul / max / min -- SIMD exp / log are out of scope,
// SURVEY.md 2b) is this kernel: dst[idx] = f(a[idx] [, b[idx]]) for every index
// of `shape`, each operand with its own element strides (0 = broadcast), alpha / beta as op parameters.
// HBM-bound: algorithmic bytes = sizeof(T) * elements * (operands read + 1 written).
// Same traversal as copy_strided (data_movement.hip): extent-1 dimensions dropped, dimensions contiguous on EVERY
// operand merged, lanes walk the innermost dimension; long rows are cut into 1024-element chunks per workgroup, short
// rows are packed several per workgroup.
#include <algorithm>
#include <type_traits>

#include "../../include/laser_hip.h"
#include "common.h"

namespace laser_hip {

struct MapArgs {
  int rank;
  int64_t shape[kMaxRank];
  int64_t sd[kMaxRank], sa[kMaxRank], sb[kMaxRank];
  int64_t inner, chunks, rows;
  int log2p;
  int op;
};

template <typename T>
__device__ __forceinline__ T map_op(int op, T a, T b, T alpha, T beta) {
  switch (op) {
    case LASER_HIP_MAP_COPY: return a;
    case LASER_HIP_MAP_FILL: return alpha;
    case LASER_HIP_MAP_NEG: return -a;
    case LASER_HIP_MAP_ABS: return a < (T)0 ? -a : a;
    case LASER_HIP_MAP_RELU: return a > (T)0 ? a : (T)0;
    case LASER_HIP_MAP_SCALE: {
#pragma clang fp contract(off)
      return alpha * a + beta;
    }
    case LASER_HIP_MAP_SQUARE: return a * a;
    case LASER_HIP_MAP_ADD: return a + b;
    case LASER_HIP_MAP_SUB: return a - b;
    case LASER_HIP_MAP_MUL: return a * b;
    case LASER_HIP_MAP_MAX: return a > b ? a : b;
    case LASER_HIP_MAP_MIN: return a < b ? a : b;
    case LASER_HIP_MAP_AXPY: {
#pragma clang fp contract(off)
      return alpha * a + b;
    }
    case LASER_HIP_MAP_AXPBY: {
#pragma clang fp contract(off)
      return alpha * a + beta * b;
    }
    default: break;
  }
  return a;
}

template <typename T, int NIN>
__global__ void __launch_bounds__(256) map_strided_kernel(T *__restrict__ dst, const T *a, const T *b, const MapArgs m, const T alpha,
                                                          const T beta) {
  const int64_t id = m.sd[m.rank - 1], ia = m.sa[m.rank - 1], ib = m.sb[m.rank - 1];
  auto outer = [&](int64_t row, int64_t &od, int64_t &oa, int64_t &ob) __attribute__((always_inline)) {
    int64_t rem = row;
    od = oa = ob = 0;
#pragma unroll
    for (int d = kMaxRank - 2; d >= 0; d--) {
      if (d < m.rank - 1) {
        const int64_t q = rem / m.shape[d], i = rem - q * m.shape[d];
        od += i * m.sd[d];
        oa += i * m.sa[d];
        ob += i * m.sb[d];
        rem = q;
      }
    }
  };
  if (m.log2p < 0) {  // long rows: one workgroup = one 1024-element chunk of one row
    const int64_t row = blockIdx.x / m.chunks, chunk = blockIdx.x - row * m.chunks;
    int64_t od, oa, ob;
    outer(row, od, oa, ob);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t i = chunk * 1024 + threadIdx.x + 256 * j;
      if (i < m.inner) {
        const T va = NIN >= 1 ? a[oa + i * ia] : (T)0, vb = NIN >= 2 ? b[ob + i * ib] : (T)0;
        dst[od + i * id] = map_op<T>(m.op, va, vb, alpha, beta);
      }
    }
  } else {  // short rows: 1024 >> log2p consecutive rows per workgroup
    const int rows_per_wg = 1024 >> m.log2p;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int slot = threadIdx.x + 256 * j;
      const int64_t row = (int64_t)blockIdx.x * rows_per_wg + (slot >> m.log2p), i = slot & ((1 << m.log2p) - 1);
      if (row < m.rows && i < m.inner) {
        int64_t od, oa, ob;
        outer(row, od, oa, ob);
        const T va = NIN >= 1 ? a[oa + i * ia] : (T)0, vb = NIN >= 2 ? b[ob + i * ib] : (T)0;
        dst[od + i * id] = map_op<T>(m.op, va, vb, alpha, beta);
      }
    }
  }
}

template <typename T>
hipError_t launch_map_strided(int op, int nin, T *dst, const int64_t *dstrides, const T *a, const int64_t *astrides, const T *b,
                              const int64_t *bstrides, const int64_t *shape, int rank, T alpha, T beta, hipStream_t s) {
  // drop extent-1 dimensions, merge dimension pairs that are contiguous on every operand
  int64_t sh[kMaxRank], sd[kMaxRank], sa[kMaxRank], sb[kMaxRank];
  int r = 0;
  int64_t total = 1;
  for (int d = 0; d < rank; d++) {
    total *= shape[d];
    if (shape[d] == 1) continue;
    const int64_t da = nin >= 1 ? astrides[d] : 0, db = nin >= 2 ? bstrides[d] : 0;
    if (r > 0 && sd[r - 1] == dstrides[d] * shape[d] && sa[r - 1] == da * shape[d] && sb[r - 1] == db * shape[d]) {
      sh[r - 1] *= shape[d];
      sd[r - 1] = dstrides[d]; sa[r - 1] = da; sb[r - 1] = db;
    } else {
      sh[r] = shape[d]; sd[r] = dstrides[d]; sa[r] = da; sb[r] = db;
      r++;
    }
  }
  if (total == 0) return hipSuccess;
  if (r == 0) { sh[0] = 1; sd[0] = 1; sa[0] = 0; sb[0] = 0; r = 1; }
  MapArgs m;
  m.rank = r;
  m.op = op;
  m.rows = 1;
  for (int d = 0; d < kMaxRank; d++) {
    m.shape[d] = d < r ? sh[d] : 1;
    m.sd[d] = d < r ? sd[d] : 0;
    m.sa[d] = d < r ? sa[d] : 0;
    m.sb[d] = d < r ? sb[d] : 0;
    if (d < r - 1) m.rows *= sh[d];
  }
  m.inner = sh[r - 1];
  int64_t blocks;
  if (m.inner < 1024 && r > 1) {
    m.log2p = 0;
    while ((1 << m.log2p) < m.inner) m.log2p++;
    m.chunks = 1;
    blocks = (m.rows + (1024 >> m.log2p) - 1) / (1024 >> m.log2p);
  } else {
    m.log2p = -1;
    m.chunks = (m.inner + 1023) / 1024;
    blocks = m.rows * m.chunks;
  }
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (nin == 0)
    hipLaunchKernelGGL((map_strided_kernel<T, 0>), dim3((unsigned)blocks), dim3(256), 0, s, dst, a, b, m, alpha, beta);
  else if (nin == 1)
    hipLaunchKernelGGL((map_strided_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, s, dst, a, b, m, alpha, beta);
  else
    hipLaunchKernelGGL((map_strided_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, s, dst, a, b, m, alpha, beta);
  return hipGetLastError();
}
#define LH_INST(T)                                                                                                     \
  template hipError_t launch_map_strided<T>(int, int, T *, const int64_t *, const T *, const int64_t *, const T *,     \
                                            const int64_t *, const int64_t *, int, T, T, hipStream_t);
LH_INST(float)
LH_INST(double)
LH_INST(int32_t)
LH_INST(int64_t)
#undef LH_INST

}  // namespace laser_hip
