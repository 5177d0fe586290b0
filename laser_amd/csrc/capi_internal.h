// laser_amd/csrc/capi_internal.h -- the few internals of capi.cpp that other host-side translation units of
// liblaser_hip.so (sharded.cpp) use.  Not part of the ABI.
#pragma once
#include <stdint.h>

namespace laser_hip {
// set the thread-local error message (printf-style) and return `code`
int api_fail(int code, const char *fmt, ...);
int api_ensure_init();
// The device the host-pointer entry points use ON THIS THREAD (-1 = the library's default device).  The sharded
// host path sets it in its per-GPU worker threads.
void api_set_thread_device(int device);
int api_thread_device();
// Tile-configuration override for the f32 launches made BY THIS THREAD (-2 = none: the process-wide setting applies).
// (Forces the compiler-scheduled kernel of that configuration: tests and tuning sweeps.)
void api_set_thread_f32_config(int cfg);
// Tile-class pin of the hand-scheduled f32 kernels for the launches made BY THIS THREAD (option "asm_tile"'s values; -2 = none).
void api_set_thread_asm_tile(int tile_class);
// sharded.cpp: ranks of the RCCL communicator the last GATHER_RCCL call used (ncclCommCount); 0 = none yet
int64_t api_shard_rccl_ranks();
// sharded.cpp: host-pointer gemm_strided cut into one row range per GPU (ndev <= 0: every visible GPU)
template <typename T>
int api_sharded_host(int ndev, int64_t M, int64_t N, int64_t K, T alpha, const T *A, int64_t rsA, int64_t csA, const T *B,
                     int64_t rsB, int64_t csB, T beta, T *C, int64_t rsC, int64_t csC);
}  // namespace laser_hip
