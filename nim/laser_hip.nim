# nim/laser_hip.nim -- the shim a Laser maintainer adds to route the GEMM hot path to an MI355X.
#
# Every exported proc below keeps the SIGNATURE of the reference proc it replaces (file:line given
# per proc) and forwards to the monomorphic C-ABI symbol of liblaser_hip.so (include/laser_hip.h).
# FFI idiom is the one the reference already uses for OpenBLAS (benchmarks/third_party/blas.nim:
# 18-23: `{.dynlib: blas, importc: "cblas_sgemm".}`) and for libjit (gemm_bench_float32.nim:199-202).
#
# NOTE: delivered as SOURCE.  The build image has no Nim compiler (probed: nim/nimble absent, no
# network), so this file has not been compiled here.  What IS checked mechanically
# (tests/test_nim_shim_cpu.py): every `importc: "..."` below names a symbol that liblaser_hip.so
# exports (nm -D) and that include/laser_hip.h declares, with the same number of parameters and
# C-compatible parameter / return types (Nim int == int64_t on amd64, cint == int, float32 == float,
# float64 == double, pointer / ptr T / cstring == pointers); every import is spelled with an explicit
# `importc: "<symbol>"` string (no identifier pasting), one declaration per C symbol.
#
# Usage inside Laser: `import laser_hip` instead of `./gemm` / `./gemm_prepacked` /
# `../swapaxes` / `./conv2d_im2col`; call sites do not change.

const laserHip* = "liblaser_hip.so"

{.pragma: lh, cdecl, dynlib: laserHip.}

# ---- C-ABI imports (Nim int == int64 on amd64, float32 == C float) ---------------------------
proc laser_hip_last_error(): cstring {.lh, importc: "laser_hip_last_error".}
proc laser_hip_abi_version*(): cint {.lh, importc: "laser_hip_abi_version".}
const laserHipAbi* = 2                       # include/laser_hip.h LASER_HIP_ABI_VERSION this shim was written against
proc laser_hip_init*(device: cint): cint {.lh, importc: "laser_hip_init".}
proc laser_hip_finalize*(): cint {.lh, importc: "laser_hip_finalize".}
proc laser_hip_device_count*(): cint {.lh, importc: "laser_hip_device_count".}
proc laser_hip_plan_f32*(M, N, K: int64, laser_order, cus: cint, out8: ptr int64): cint {.lh, importc: "laser_hip_plan_f32".}   # diagnostics: kernel + launch plan for a device of `cus` CUs
proc laser_hip_set_float_mode*(mode: cint): cint {.lh, importc: "laser_hip_set_float_mode".}   # 0 = Laser order (default), 1 = fast
# every tuning / A-B switch by name (include/laser_hip.h lists them), and the read-only diagnostics of the last launch
proc laser_hip_set_option(name: cstring, value: cint): cint {.lh, importc: "laser_hip_set_option".}
proc laser_hip_get_option(name: cstring, value: ptr int): cint {.lh, importc: "laser_hip_get_option".}

proc laserHipCheckAbi*() =
  ## call once at start-up: a library with another ABI number takes other argument types for the same symbol names
  doAssert laser_hip_abi_version() == laserHipAbi, "liblaser_hip.so has ABI " & $laser_hip_abi_version() & ", this shim expects " & $laserHipAbi

template check(rc: cint) =
  # the reference procs return void and doAssert on precondition violations
  # (gemm_prepacked.nim:125,208; conv2d_im2col.nim:109); keep that contract
  let code = rc
  doAssert code == 0, "laser_hip error " & $code & ": " & $laser_hip_last_error()

# gemm_strided, host pointers (blocking, Laser's call semantics)
proc laser_hip_gemm_strided_f32(M, N, K: int, alpha: float32, A: ptr float32, rsA, csA: int, B: ptr float32, rsB, csB: int, beta: float32, C: ptr float32, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_strided_f32".}
proc laser_hip_gemm_strided_f64(M, N, K: int, alpha: float64, A: ptr float64, rsA, csA: int, B: ptr float64, rsB, csB: int, beta: float64, C: ptr float64, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_strided_f64".}
proc laser_hip_gemm_strided_i32(M, N, K: int, alpha: int32, A: ptr int32, rsA, csA: int, B: ptr int32, rsB, csB: int, beta: int32, C: ptr int32, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_strided_i32".}
proc laser_hip_gemm_strided_i64(M, N, K: int, alpha: int64, A: ptr int64, rsA, csA: int, B: ptr int64, rsB, csB: int, beta: int64, C: ptr int64, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_strided_i64".}
# gemm_strided, device pointers + stream (asynchronous; operands resident in HBM)
proc laser_hip_gemm_strided_f32_dev(M, N, K: int, alpha: float32, A: pointer, rsA, csA: int, B: pointer, rsB, csB: int, beta: float32, C: pointer, rsC, csC: int, stream: pointer): cint {.lh, importc: "laser_hip_gemm_strided_f32_dev".}
proc laser_hip_gemm_strided_f64_dev(M, N, K: int, alpha: float64, A: pointer, rsA, csA: int, B: pointer, rsB, csB: int, beta: float64, C: pointer, rsC, csC: int, stream: pointer): cint {.lh, importc: "laser_hip_gemm_strided_f64_dev".}
proc laser_hip_gemm_strided_i32_dev(M, N, K: int, alpha: int32, A: pointer, rsA, csA: int, B: pointer, rsB, csB: int, beta: int32, C: pointer, rsC, csC: int, stream: pointer): cint {.lh, importc: "laser_hip_gemm_strided_i32_dev".}
proc laser_hip_gemm_strided_i64_dev(M, N, K: int, alpha: int64, A: pointer, rsA, csA: int, B: pointer, rsB, csB: int, beta: int64, C: pointer, rsC, csC: int, stream: pointer): cint {.lh, importc: "laser_hip_gemm_strided_i64_dev".}

# pre-packed GEMM
proc laser_hip_gemm_prepackA_mem_required_f32(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackA_mem_required_f32".}
proc laser_hip_gemm_prepackA_mem_required_f64(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackA_mem_required_f64".}
proc laser_hip_gemm_prepackA_mem_required_i32(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackA_mem_required_i32".}
proc laser_hip_gemm_prepackA_mem_required_i64(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackA_mem_required_i64".}
proc laser_hip_gemm_prepackB_mem_required_f32(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackB_mem_required_f32".}
proc laser_hip_gemm_prepackB_mem_required_f64(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackB_mem_required_f64".}
proc laser_hip_gemm_prepackB_mem_required_i32(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackB_mem_required_i32".}
proc laser_hip_gemm_prepackB_mem_required_i64(M, N, K: int): int {.lh, importc: "laser_hip_gemm_prepackB_mem_required_i64".}
proc laser_hip_gemm_prepackA_f32(dst: pointer, M, N, K: int, A: ptr float32, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackA_f32".}
proc laser_hip_gemm_prepackA_f64(dst: pointer, M, N, K: int, A: ptr float64, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackA_f64".}
proc laser_hip_gemm_prepackA_i32(dst: pointer, M, N, K: int, A: ptr int32, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackA_i32".}
proc laser_hip_gemm_prepackA_i64(dst: pointer, M, N, K: int, A: ptr int64, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackA_i64".}
proc laser_hip_gemm_prepackB_f32(dst: pointer, M, N, K: int, B: ptr float32, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackB_f32".}
proc laser_hip_gemm_prepackB_f64(dst: pointer, M, N, K: int, B: ptr float64, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackB_f64".}
proc laser_hip_gemm_prepackB_i32(dst: pointer, M, N, K: int, B: ptr int32, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackB_i32".}
proc laser_hip_gemm_prepackB_i64(dst: pointer, M, N, K: int, B: ptr int64, rs, cs: int): cint {.lh, importc: "laser_hip_gemm_prepackB_i64".}
proc laser_hip_gemm_packed_f32(M, N, K: int, alpha: float32, pA, pB: pointer, beta: float32, C: ptr float32, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_packed_f32".}
proc laser_hip_gemm_packed_f64(M, N, K: int, alpha: float64, pA, pB: pointer, beta: float64, C: ptr float64, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_packed_f64".}
proc laser_hip_gemm_packed_i32(M, N, K: int, alpha: int32, pA, pB: pointer, beta: int32, C: ptr int32, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_packed_i32".}
proc laser_hip_gemm_packed_i64(M, N, K: int, alpha: int64, pA, pB: pointer, beta: int64, C: ptr int64, rsC, csC: int): cint {.lh, importc: "laser_hip_gemm_packed_i64".}
proc laser_hip_gemm_prepack_release*(packed: pointer): cint {.lh, importc: "laser_hip_gemm_prepack_release".}

# physical transposes
proc laser_hip_transpose2d_copy_b32(dst, src: pointer, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_copy_b32".}
proc laser_hip_transpose2d_copy_b64(dst, src: pointer, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_copy_b64".}
proc laser_hip_transpose2d_batched_b32(dst, src: pointer, N, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_batched_b32".}
proc laser_hip_transpose2d_batched_b64(dst, src: pointer, N, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_batched_b64".}
proc laser_hip_transpose2d_copy_b16(dst, src: pointer, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_copy_b16".}
proc laser_hip_transpose2d_copy_b8(dst, src: pointer, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_copy_b8".}
proc laser_hip_transpose2d_batched_b16(dst, src: pointer, N, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_batched_b16".}
proc laser_hip_transpose2d_batched_b8(dst, src: pointer, N, NR, NC: int): cint {.lh, importc: "laser_hip_transpose2d_batched_b8".}

# im2col + GEMM convolution, cblas-shaped gemm
proc laser_hip_im2col_workspace_size(iN, iC, iH, iW, cOut, cIn, kH, kW, pH, pW, sH, sW: int): int {.lh, importc: "laser_hip_im2col_workspace_size".}
proc laser_hip_im2col_f32(ws: ptr float32, oH, oW: int, input: ptr float32, iC, iH, iW, kH, kW, pH, pW, sH, sW: int): cint {.lh, importc: "laser_hip_im2col_f32".}
proc laser_hip_im2col_f64(ws: ptr float64, oH, oW: int, input: ptr float64, iC, iH, iW, kH, kW, pH, pW, sH, sW: int): cint {.lh, importc: "laser_hip_im2col_f64".}
proc laser_hip_conv2d_im2col_f32(output, input: ptr float32, iN, iC, iH, iW: int, kernel: ptr float32, cOut, cIn, kH, kW, pH, pW, sH, sW: int, ws: ptr float32): cint {.lh, importc: "laser_hip_conv2d_im2col_f32".}
proc laser_hip_cblas_sgemm(order, tA, tB: cint, M, N, K: int, alpha: float32, A: ptr float32, lda: int, B: ptr float32, ldb: int, beta: float32, C: ptr float32, ldc: int): cint {.lh, importc: "laser_hip_cblas_sgemm".}
proc laser_hip_cblas_dgemm(order, tA, tB: cint, M, N, K: int, alpha: float64, A: ptr float64, lda: int, B: ptr float64, ldb: int, beta: float64, C: ptr float64, ldc: int): cint {.lh, importc: "laser_hip_cblas_dgemm".}

# fused epilogue
proc laser_hip_gemm_strided_ex_f32(M, N, K: int, alpha: float32, A: ptr float32, rsA, csA: int, B: ptr float32, rsB, csB: int, beta: float32, C: ptr float32, rsC, csC: int, bias: ptr float32, rsBias, csBias: int, activation: cint): cint {.lh, importc: "laser_hip_gemm_strided_ex_f32".}
proc laser_hip_gemm_strided_ex_f64(M, N, K: int, alpha: float64, A: ptr float64, rsA, csA: int, B: ptr float64, rsB, csB: int, beta: float64, C: ptr float64, rsC, csC: int, bias: ptr float64, rsBias, csBias: int, activation: cint): cint {.lh, importc: "laser_hip_gemm_strided_ex_f64".}

# device tensor storage
proc laser_hip_storage_alloc(d: ptr pointer, bytes: int): cint {.lh, importc: "laser_hip_storage_alloc".}
proc laser_hip_storage_free(d: pointer): cint {.lh, importc: "laser_hip_storage_free".}
proc laser_hip_storage_upload(d, host: pointer, bytes: int): cint {.lh, importc: "laser_hip_storage_upload".}
proc laser_hip_storage_download(host, d: pointer, bytes: int): cint {.lh, importc: "laser_hip_storage_download".}
proc laser_hip_storage_set_zero(d: pointer, bytes: int, stream: pointer): cint {.lh, importc: "laser_hip_storage_set_zero".}
proc laser_hip_copy_strided_b32_dev(dst: pointer, dstStrides: ptr int, src: pointer, srcStrides: ptr int, shape: ptr int, rank: cint, stream: pointer): cint {.lh, importc: "laser_hip_copy_strided_b32_dev".}
proc laser_hip_copy_strided_b64_dev(dst: pointer, dstStrides: ptr int, src: pointer, srcStrides: ptr int, shape: ptr int, rank: cint, stream: pointer): cint {.lh, importc: "laser_hip_copy_strided_b64_dev".}

# ---- gemm_strided -- laser/primitives/matrix_multiplication/gemm.nim:184-193 --------------------
# Element types dispatched by the reference: float32, float64, int32 / uint32 (one branch, gemm.nim:239),
# int64.  Integer arithmetic wraps modulo 2^n, so uint32 is the int32 entry point on the same bits.
proc gemm_strided*[T: SomeNumber](
      M, N, K: int,
      alpha: T,
      A: ptr T,
      rowStrideA, colStrideA: int,
      B: ptr T,
      rowStrideB, colStrideB: int,
      beta: T,
      C: ptr T,
      rowStrideC, colStrideC: int) =
  when T is float32:
    check laser_hip_gemm_strided_f32(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C, rowStrideC, colStrideC)
  elif T is float64:
    check laser_hip_gemm_strided_f64(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C, rowStrideC, colStrideC)
  elif T is int32:
    check laser_hip_gemm_strided_i32(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta, C, rowStrideC, colStrideC)
  elif T is uint32:
    check laser_hip_gemm_strided_i32(M, N, K, cast[int32](alpha), cast[ptr int32](A), rowStrideA, colStrideA,
                                     cast[ptr int32](B), rowStrideB, colStrideB, cast[int32](beta),
                                     cast[ptr int32](C), rowStrideC, colStrideC)
  elif T is int64 or T is int:
    check laser_hip_gemm_strided_i64(M, N, K, cast[int64](alpha), cast[ptr int64](A), rowStrideA, colStrideA,
                                     cast[ptr int64](B), rowStrideB, colStrideB, cast[int64](beta),
                                     cast[ptr int64](C), rowStrideC, colStrideC)
  else:
    {.error: "laser_hip: unsupported element type " & $T.}

# ---- fused epilogue -- planned in the reference (README.md:238-242; TODO gemm.nim:196) -------------
# C = act(alpha*A*B + beta*C + bias); bias is a strided M x N view whose strides may be 0.
type Activation* = enum
  actNone = 0, actRelu = 1, actTanh = 2, actSigmoid = 3
# fused PROLOGUE -- "fuse operations before the matrix multiplication kernel, during the prepacking" (README.md:243-244):
# the product is taken of relu(A) and / or relu(B); the bits travel in the activation argument (LASER_HIP_PRE_RELU_A / _B)
type Prologue* = enum
  preNone = 0, preReluA = 0x100, preReluB = 0x200, preReluAB = 0x300

proc gemm_strided_fused*[T: float32 or float64](
      M, N, K: int, alpha: T,
      A: ptr T, rowStrideA, colStrideA: int,
      B: ptr T, rowStrideB, colStrideB: int,
      beta: T,
      C: ptr T, rowStrideC, colStrideC: int,
      bias: ptr T, rowStrideBias, colStrideBias: int,
      activation = actNone, prologue = preNone) =
  let code = cint(ord(activation) or ord(prologue))
  when T is float32:
    check laser_hip_gemm_strided_ex_f32(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta,
                                        C, rowStrideC, colStrideC, bias, rowStrideBias, colStrideBias, code)
  else:
    check laser_hip_gemm_strided_ex_f64(M, N, K, alpha, A, rowStrideA, colStrideA, B, rowStrideB, colStrideB, beta,
                                        C, rowStrideC, colStrideC, bias, rowStrideBias, colStrideBias, code)

# ---- pre-packed GEMM -- gemm_prepacked.nim:76-292 -----------------------------------------------
proc gemm_prepackB_mem_required*(T: type, M, N, K: int): int =
  when T is float32: laser_hip_gemm_prepackB_mem_required_f32(M, N, K)
  elif T is float64: laser_hip_gemm_prepackB_mem_required_f64(M, N, K)
  elif T is int32 or T is uint32: laser_hip_gemm_prepackB_mem_required_i32(M, N, K)
  else: laser_hip_gemm_prepackB_mem_required_i64(M, N, K)

proc gemm_prepackA_mem_required*(T: type, M, N, K: int): int =
  when T is float32: laser_hip_gemm_prepackA_mem_required_f32(M, N, K)
  elif T is float64: laser_hip_gemm_prepackA_mem_required_f64(M, N, K)
  elif T is int32 or T is uint32: laser_hip_gemm_prepackA_mem_required_i32(M, N, K)
  else: laser_hip_gemm_prepackA_mem_required_i64(M, N, K)

proc gemm_prepackB*[T](dst_packedB: ptr (T or UncheckedArray[T]), M, N, K: int,
                       src_B: ptr T, rowStrideB, colStrideB: int) =
  when T is float32: check laser_hip_gemm_prepackB_f32(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)
  elif T is float64: check laser_hip_gemm_prepackB_f64(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)
  elif T is int32: check laser_hip_gemm_prepackB_i32(dst_packedB, M, N, K, src_B, rowStrideB, colStrideB)
  else: check laser_hip_gemm_prepackB_i64(dst_packedB, M, N, K, cast[ptr int64](src_B), rowStrideB, colStrideB)

proc gemm_prepackA*[T](dst_packedA: ptr (T or UncheckedArray[T]), M, N, K: int,
                       src_A: ptr T, rowStrideA, colStrideA: int) =
  when T is float32: check laser_hip_gemm_prepackA_f32(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)
  elif T is float64: check laser_hip_gemm_prepackA_f64(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)
  elif T is int32: check laser_hip_gemm_prepackA_i32(dst_packedA, M, N, K, src_A, rowStrideA, colStrideA)
  else: check laser_hip_gemm_prepackA_i64(dst_packedA, M, N, K, cast[ptr int64](src_A), rowStrideA, colStrideA)

proc gemm_packed*[T: SomeNumber](M, N, K: int, alpha: T,
      packedA: ptr (T or UncheckedArray[T]), packedB: ptr (T or UncheckedArray[T]),
      beta: T, C: ptr (T or UncheckedArray[T]), rowStrideC, colStrideC: int) =
  when T is float32: check laser_hip_gemm_packed_f32(M, N, K, alpha, packedA, packedB, beta, cast[ptr float32](C), rowStrideC, colStrideC)
  elif T is float64: check laser_hip_gemm_packed_f64(M, N, K, alpha, packedA, packedB, beta, cast[ptr float64](C), rowStrideC, colStrideC)
  elif T is int32: check laser_hip_gemm_packed_i32(M, N, K, alpha, packedA, packedB, beta, cast[ptr int32](C), rowStrideC, colStrideC)
  else: check laser_hip_gemm_packed_i64(M, N, K, cast[int64](alpha), packedA, packedB, cast[int64](beta), cast[ptr int64](C), rowStrideC, colStrideC)

# ---- physical transposes -- laser/primitives/swapaxes.nim:16-112 ---------------------------------
proc transpose2D_copy*[T](dst, src: ptr (T or UncheckedArray[T]), NR, NC: Natural) =
  when sizeof(T) == 4: check laser_hip_transpose2d_copy_b32(dst, src, NR, NC)
  elif sizeof(T) == 8: check laser_hip_transpose2d_copy_b64(dst, src, NR, NC)
  elif sizeof(T) == 2: check laser_hip_transpose2d_copy_b16(dst, src, NR, NC)
  elif sizeof(T) == 1: check laser_hip_transpose2d_copy_b8(dst, src, NR, NC)
  else: {.error: "laser_hip transposes support 1-, 2-, 4- and 8-byte elements".}

proc transpose2D_batched*[T](dst, src: ptr (T or UncheckedArray[T]), N, NR, NC: Natural) =
  when sizeof(T) == 4: check laser_hip_transpose2d_batched_b32(dst, src, N, NR, NC)
  elif sizeof(T) == 8: check laser_hip_transpose2d_batched_b64(dst, src, N, NR, NC)
  elif sizeof(T) == 2: check laser_hip_transpose2d_batched_b16(dst, src, N, NR, NC)
  elif sizeof(T) == 1: check laser_hip_transpose2d_batched_b8(dst, src, N, NR, NC)
  else: {.error: "laser_hip transposes support 1-, 2-, 4- and 8-byte elements".}

proc nchw2nhwc*[T](dst_nhwc, src_nchw: ptr (T or UncheckedArray[T]), N, C, H, W: Natural) {.inline.} =
  transpose2D_batched(dst_nhwc, src_nchw, N, C, H*W)       # swapaxes.nim:98

proc nhwc2nchw*[T](dst_nchw, src_nhwc: ptr (T or UncheckedArray[T]), N, C, H, W: Natural) {.inline.} =
  transpose2D_batched(dst_nchw, src_nhwc, N, H*W, C)       # swapaxes.nim:112

# ---- im2col + GEMM convolution -- benchmarks/convolution/conv2d_im2col.nim -----------------------
# The convolution benchmark's own types, conv2d_common.nim:6-13.  Its `Tensor[T]` is a plain `seq[T]`
# (conv2d_common.nim:13: `Tensor*[T] = seq[T]`) -- NOT laser/tensor's Tensor -- which is why
# conv2d_bench.nim:92-102 can hand seqs to conv2d_im2col.  A maintainer who keeps
# `import ./conv2d_common` drops the five declarations below.
type
  TensorShape* = tuple[n, c, h, w: int]
  KernelShape* = tuple[c_out, c_in, kH, kW: int]
  Padding* = tuple[h, w: int]
  Strides* = tuple[h, w: int]
  Tensor*[T] = seq[T]

proc im2col_workspace_size*(ishape: TensorShape, kshape: KernelShape, padding: Padding, strides: Strides): int =
  laser_hip_im2col_workspace_size(ishape.n, ishape.c, ishape.h, ishape.w, kshape.c_out, kshape.c_in,
                                  kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w)

# conv2d_im2col.nim:42-50: generic in T (pure data movement: by element size -- int32 rides on the float32 entry point, int64 on the float64 one)
proc im2col*[T](pworkspace: ptr T, oshape: TensorShape, pinput: ptr UncheckedArray[T],
                ishape: TensorShape, kshape: KernelShape, padding: Padding, strides: Strides) =
  when sizeof(T) == 4:
    check laser_hip_im2col_f32(cast[ptr float32](pworkspace), oshape.h, oshape.w, cast[ptr float32](pinput), ishape.c, ishape.h,
                               ishape.w, kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w)
  elif sizeof(T) == 8:
    check laser_hip_im2col_f64(cast[ptr float64](pworkspace), oshape.h, oshape.w, cast[ptr float64](pinput), ishape.c, ishape.h,
                               ishape.w, kshape.kH, kshape.kW, padding.h, padding.w, strides.h, strides.w)
  else: {.error: "laser_hip im2col supports 4- and 8-byte elements".}

# conv2d_im2col.nim:90-100, parameter for parameter
proc conv2d_im2col*(
    output: var Tensor[float32], # Output tensor
    oshape: TensorShape,         # Shape of output
    input: Tensor[float32],      # Input tensor
    ishape: TensorShape,         # Shape of input
    kernel: Tensor[float32],     # Convolution filter
    kshape: KernelShape,         # kernel shape
    padding: Padding,            # Padding
    strides: Strides,            # Strides
    pworkspace: ptr float32      # Workspace buffer, can be reused between batches
  ) =
  assert oshape.c == kshape.c_out                        # conv2d_im2col.nim:109
  check laser_hip_conv2d_im2col_f32(output[0].addr, input[0].unsafeAddr, ishape.n, ishape.c, ishape.h, ishape.w,
                                    kernel[0].unsafeAddr, kshape.c_out, kshape.c_in, kshape.kH, kshape.kW,
                                    padding.h, padding.w, strides.h, strides.w, pworkspace)

# ---- cblas-shaped gemm -- benchmarks/third_party/blas.nim:12-23 ----------------------------------
type
  TransposeType* {.size: sizeof(cint).} = enum
    noTranspose = 111, transpose = 112, conjTranspose = 113
  OrderType* {.size: sizeof(cint).} = enum
    rowMajor = 101, colMajor = 102

# blas.nim:18-20 (cblas_sgemm)
proc gemm*(ORDER: OrderType, TRANSA, TRANSB: TransposeType, M, N, K: int, ALPHA: float32,
           A: ptr float32, LDA: int, B: ptr float32, LDB: int, BETA: float32, C: ptr float32, LDC: int) =
  check laser_hip_cblas_sgemm(cint(ORDER), cint(TRANSA), cint(TRANSB), M, N, K, ALPHA, A, LDA, B, LDB, BETA, C, LDC)

# blas.nim:21-23 (cblas_dgemm)
proc gemm*(ORDER: OrderType, TRANSA, TRANSB: TransposeType, M, N, K: int, ALPHA: float64,
           A: ptr float64, LDA: int, B: ptr float64, LDB: int, BETA: float64, C: ptr float64, LDC: int) =
  check laser_hip_cblas_dgemm(cint(ORDER), cint(TRANSA), cint(TRANSB), M, N, K, ALPHA, A, LDA, B, LDB, BETA, C, LDC)

# ---- RawTensor / forEach surface ------------------------------------------------------------------
# Nothing to replace for HOST tensors: `Tensor[T]`, `unsafe_raw_data`, `forEach` stay Laser's own
# (laser/tensor/datatypes.nim:13-30, laser/strided_iteration/foreach.nim:192-264).  A caller does
#   gemm_strided(M, N, K, 1'f32, a.unsafe_raw_data, a.strides[0], a.strides[1], ...)
# exactly as before; only raw pointers and element strides cross the ABI.

# ---- device-resident storage: HipStorage, the twin of CpuStorage ---------------------------------
# (SURVEY.md section 8f rank 3.)  A tensor keeps its shape / strides / offset; what changes is where
# `storage.raw_buffer` lives.  GUARD: a device address must never reach Laser's host `forEach`, whose
# duck-typing contract is `rank, size, shape, strides, unsafe_raw_data` (foreach.nim:7-29,
# laser/strided_iteration/README.md:101-107) -- it would dereference the pointer on the CPU.  So
#   * the device pointer type is `DevicePtr[T]`, a `distinct pointer` with NO `[]` and no conversion
#     to `ptr T`: it cannot be passed to gemm_strided / transpose2D_copy (host entry points) by accident;
#   * HipStorage has NO `unsafe_raw_data`; its accessor is `unsafe_device_data`, so `forEach x in t`
#     on a HipStorage-backed tensor fails to COMPILE (undeclared identifier in forEach's expansion);
#   * elementwise work on device tensors goes through `mapStrided` below
#     (laser_hip_map_strided_*_dev), the device-side twin of forEach's strided iteration.
type
  DevicePtr*[T] = distinct pointer       # an HBM address: not dereferenceable on the host
  HipStorage*{.shallow.}[T] = ref object # same layout idea as CpuStorage, datatypes.nim:24-30
    raw_buffer*: DevicePtr[T]
    memalloc*: pointer
    memowner*: bool

func unsafe_device_data*[T](s: HipStorage[T]): DevicePtr[T] {.inline.} = s.raw_buffer
func offsetBy*[T](p: DevicePtr[T], elements: int): DevicePtr[T] {.inline.} =
  DevicePtr[T](cast[pointer](cast[uint](pointer(p)) + uint(elements * sizeof(T))))

proc finalizer[T](storage: HipStorage[T]) =
  if storage.memowner and not storage.memalloc.isNil:
    discard laser_hip_storage_free(storage.memalloc)

proc allocHipStorage*[T](storage: var HipStorage[T], size: int) =
  ## allocCpuStorage's twin (allocator.nim:17-29): `size` elements, aligned >= LASER_MEM_ALIGN, zero-filled.
  new(storage, finalizer[T])
  check laser_hip_storage_alloc(storage.memalloc.addr, sizeof(T) * size)
  storage.memowner = true
  storage.raw_buffer = DevicePtr[T](storage.memalloc)

proc copyFromRaw*[T](dst: HipStorage[T], buffer: ptr T, len: Natural) =
  ## initialization.nim:112-128 for a device storage (host buffer -> HBM).
  check laser_hip_storage_upload(pointer(dst.raw_buffer), buffer, sizeof(T) * len)

proc copyToRaw*[T](buffer: ptr T, src: HipStorage[T], len: Natural) =
  check laser_hip_storage_download(buffer, pointer(src.raw_buffer), sizeof(T) * len)

proc setZero*[T](s: HipStorage[T], len: Natural) =
  check laser_hip_storage_set_zero(pointer(s.raw_buffer), sizeof(T) * len, nil)

proc copyStrided*[T](dst: DevicePtr[T], dstStrides: openarray[int], src: DevicePtr[T], srcStrides: openarray[int],
                     shape: openarray[int]) =
  ## `forEachStrided d in dst, s in src: d = s` on device buffers (deepCopy / copyFrom of views).
  assert shape.len == dstStrides.len and shape.len == srcStrides.len and shape.len <= 6   # LASER_MAXRANK
  when sizeof(T) == 4:
    check laser_hip_copy_strided_b32_dev(pointer(dst), dstStrides[0].unsafeAddr, pointer(src), srcStrides[0].unsafeAddr,
                                         shape[0].unsafeAddr, cint(shape.len), nil)
  else:
    check laser_hip_copy_strided_b64_dev(pointer(dst), dstStrides[0].unsafeAddr, pointer(src), srcStrides[0].unsafeAddr,
                                         shape[0].unsafeAddr, cint(shape.len), nil)

# ---- mapStrided: the device twin of forEach / forEachStrided (foreach.nim:192-264) ----------------
# dst[idx] = f(a[idx]) / f(a[idx], b[idx]) over rank <= 6 (LASER_MAXRANK) strided views, strides in elements, 0 = broadcast.
type MapOp* = enum
  mapCopy = 0, mapFill = 1, mapNeg = 2, mapAbs = 3, mapRelu = 4, mapScale = 5, mapSquare = 6,
  mapAdd = 32, mapSub = 33, mapMul = 34, mapMax = 36, mapMin = 37, mapAxpy = 38, mapAxpby = 39

proc laser_hip_map_strided_unary_f32_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: float32, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_unary_f32_dev".}
proc laser_hip_map_strided_unary_f64_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: float64, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_unary_f64_dev".}
proc laser_hip_map_strided_unary_i32_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: int32, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_unary_i32_dev".}
proc laser_hip_map_strided_unary_i64_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: int64, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_unary_i64_dev".}
proc laser_hip_map_strided_binary_f32_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, b: pointer, bStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: float32, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_binary_f32_dev".}
proc laser_hip_map_strided_binary_f64_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, b: pointer, bStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: float64, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_binary_f64_dev".}
proc laser_hip_map_strided_binary_i32_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, b: pointer, bStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: int32, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_binary_i32_dev".}
proc laser_hip_map_strided_binary_i64_dev(op: cint, dst: pointer, dstStrides: ptr int, a: pointer, aStrides: ptr int, b: pointer, bStrides: ptr int, shape: ptr int, rank: cint, alpha, beta: int64, stream: pointer): cint {.lh, importc: "laser_hip_map_strided_binary_i64_dev".}

proc mapStrided*[T: float32 or float64 or int32 or int64](op: MapOp, dst: DevicePtr[T], dstStrides: openarray[int],
                 a: DevicePtr[T], aStrides: openarray[int], shape: openarray[int],
                 alpha = T(1), beta = T(0), stream: pointer = nil) =
  ## `forEachStrided d in dst, x in a: d = f(x)` on device buffers (fill: `a` may be a nil DevicePtr).
  assert shape.len == dstStrides.len and shape.len == aStrides.len and shape.len <= 6 and ord(op) < 32
  when T is float32:
    check laser_hip_map_strided_unary_f32_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)
  elif T is float64:
    check laser_hip_map_strided_unary_f64_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)
  elif T is int32:
    check laser_hip_map_strided_unary_i32_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)
  else:
    check laser_hip_map_strided_unary_i64_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)

proc mapStrided*[T: float32 or float64 or int32 or int64](op: MapOp, dst: DevicePtr[T], dstStrides: openarray[int],
                 a: DevicePtr[T], aStrides: openarray[int], b: DevicePtr[T], bStrides: openarray[int],
                 shape: openarray[int], alpha = T(1), beta = T(1), stream: pointer = nil) =
  ## `forEachStrided d in dst, x in a, y in b: d = f(x, y)` on device buffers.
  assert shape.len == dstStrides.len and shape.len == aStrides.len and shape.len == bStrides.len and shape.len <= 6 and ord(op) >= 32
  when T is float32:
    check laser_hip_map_strided_binary_f32_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, pointer(b), bStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)
  elif T is float64:
    check laser_hip_map_strided_binary_f64_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, pointer(b), bStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)
  elif T is int32:
    check laser_hip_map_strided_binary_i32_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, pointer(b), bStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)
  else:
    check laser_hip_map_strided_binary_i64_dev(cint(ord(op)), pointer(dst), dstStrides[0].unsafeAddr, pointer(a), aStrides[0].unsafeAddr, pointer(b), bStrides[0].unsafeAddr, shape[0].unsafeAddr, cint(shape.len), alpha, beta, stream)

# ---- the whole node behind an unchanged gemm_strided call, and pinned host memory ------------------------------------
# Laser parallelises one gemm_strided call over OpenMP threads by ic row blocks (gemm.nim:160-176); here the same row blocks
# go to the GPUs of the node: after `laserHipShardDevices(0)` (0 = every visible GPU, n = that many, 1 = off) the plain
# host-pointer gemm_strided above cuts large problems into one row range per GPU inside the library -- call sites unchanged.
proc laser_hip_set_shard_devices(ndev: cint): cint {.lh, importc: "laser_hip_set_shard_devices".}
proc laser_hip_get_shard_devices(): cint {.lh, importc: "laser_hip_get_shard_devices".}
proc laser_hip_host_alloc(hostPtr: ptr pointer, bytes: int): cint {.lh, importc: "laser_hip_host_alloc".}
proc laser_hip_host_free(hostPtr: pointer): cint {.lh, importc: "laser_hip_host_free".}
proc laser_hip_host_register(hostPtr: pointer, bytes: int): cint {.lh, importc: "laser_hip_host_register".}
proc laser_hip_host_unregister(hostPtr: pointer): cint {.lh, importc: "laser_hip_host_unregister".}

proc laserHipSetOption*(name: string, value: int) = check laser_hip_set_option(cstring(name), cint(value))
proc laserHipGetOption*(name: string): int =
  check laser_hip_get_option(cstring(name), result.addr)
proc laserHipShardDevices*(ndev: int) = check laser_hip_set_shard_devices(cint(ndev))
proc laserHipShardDevices*(): int = int(laser_hip_get_shard_devices())

# ---- one gemm_strided call over every GPU, operands resident in HBM, C all-gathered over xGMI ------------------------
# The device-resident form of the same partition: rows of C are dealt block-cyclically (`shardPlan`), device slot g holds
# its row panels of A, all of B, and the FULL C; with a gather mode every device ends up with all of C -- sub-panel s is
# sent (peer copies on every xGMI link at once, or RCCL's ncclAllGather) while sub-panel s+1 multiplies.  No K split, so
# every element is computed exactly as on one GPU (gemm.nim:160-176 partitions M the same way across threads).
type
  ShardGather* = enum
    gatherNone = 0, gatherPeer = 1, gatherRccl = 2
  ShardPlan* = object
    rowsPerPanel*: int   ## rows of one (sub-panel, device) block; a multiple of 256 when M allows
    panelsPerDev*: int   ## sub-panels per device actually used
    paddedM*: int        ## rows the per-device C buffers must hold for gatherRccl (whole panels)
const shardPinTile* = 1  ## flags: 128x128 tiles for the local products (RCCL's kernels hold CUs meanwhile)

proc laser_hip_shard_plan(M: int, ndev, panelsPerDev: cint, rowsPerPanel: ptr int, panelsPerDevUsed: ptr cint, paddedM: ptr int): cint {.lh, importc: "laser_hip_shard_plan".}
proc laser_hip_gemm_strided_f32_sharded_dev(ndev: cint, devices: ptr cint, M, N, K: int, alpha: float32, dA: ptr pointer, rsA, csA: int, dB: ptr pointer, rsB, csB: int, beta: float32, dC: ptr pointer, rsC: int, panelsPerDev, gather, flags: cint): cint {.lh, importc: "laser_hip_gemm_strided_f32_sharded_dev".}
proc laser_hip_gemm_strided_f64_sharded_dev(ndev: cint, devices: ptr cint, M, N, K: int, alpha: float64, dA: ptr pointer, rsA, csA: int, dB: ptr pointer, rsB, csB: int, beta: float64, dC: ptr pointer, rsC: int, panelsPerDev, gather, flags: cint): cint {.lh, importc: "laser_hip_gemm_strided_f64_sharded_dev".}
proc laser_hip_gemm_strided_i32_sharded_dev(ndev: cint, devices: ptr cint, M, N, K: int, alpha: int32, dA: ptr pointer, rsA, csA: int, dB: ptr pointer, rsB, csB: int, beta: int32, dC: ptr pointer, rsC: int, panelsPerDev, gather, flags: cint): cint {.lh, importc: "laser_hip_gemm_strided_i32_sharded_dev".}
proc laser_hip_gemm_strided_i64_sharded_dev(ndev: cint, devices: ptr cint, M, N, K: int, alpha: int64, dA: ptr pointer, rsA, csA: int, dB: ptr pointer, rsB, csB: int, beta: int64, dC: ptr pointer, rsC: int, panelsPerDev, gather, flags: cint): cint {.lh, importc: "laser_hip_gemm_strided_i64_sharded_dev".}

proc shardPlan*(M, ndev: int, panelsPerDev = 4): ShardPlan =
  var ppd: cint
  check laser_hip_shard_plan(M, cint(ndev), cint(panelsPerDev), result.rowsPerPanel.addr, ppd.addr, result.paddedM.addr)
  result.panelsPerDev = int(ppd)

# (The device-resident sharded call has no stream parameter: it works on the library's own streams, so every operand must be COMPLETE in
# memory when it is made -- a caller that fills its device tensors asynchronously synchronises that stream first.  flags = 1
# (LASER_HIP_SHARD_PIN_TILE): the hand-scheduled 128x128x16 assembly tile for the local float32 products, beside RCCL's kernels.)
proc gemm_strided_sharded*[T: float32 or float64 or int32 or int64](
      devices: openarray[cint],                 ## HIP ordinals, one per device slot
      M, N, K: int, alpha: T,
      A_panels: openarray[DevicePtr[T]],        ## per slot: its row panels of A, stacked in local order
      rowStrideA, colStrideA: int,
      B: openarray[DevicePtr[T]],               ## per slot: B (replicated)
      rowStrideB, colStrideB: int,
      beta: T,
      C: openarray[DevicePtr[T]],               ## per slot: the full row-major C
      rowStrideC: int,
      panelsPerDev = 4, gather = gatherPeer, flags = 0) =
  ## Same arithmetic as `gemm_strided` (bit-identical for every device count); synchronous.
  assert A_panels.len == devices.len and B.len == devices.len and C.len == devices.len
  let n = cint(devices.len)
  let dv = devices[0].unsafeAddr
  let a = cast[ptr pointer](A_panels[0].unsafeAddr)
  let b = cast[ptr pointer](B[0].unsafeAddr)
  let c = cast[ptr pointer](C[0].unsafeAddr)
  when T is float32:
    check laser_hip_gemm_strided_f32_sharded_dev(n, dv, M, N, K, alpha, a, rowStrideA, colStrideA, b, rowStrideB, colStrideB, beta, c, rowStrideC, cint(panelsPerDev), cint(ord(gather)), cint(flags))
  elif T is float64:
    check laser_hip_gemm_strided_f64_sharded_dev(n, dv, M, N, K, alpha, a, rowStrideA, colStrideA, b, rowStrideB, colStrideB, beta, c, rowStrideC, cint(panelsPerDev), cint(ord(gather)), cint(flags))
  elif T is int32:
    check laser_hip_gemm_strided_i32_sharded_dev(n, dv, M, N, K, alpha, a, rowStrideA, colStrideA, b, rowStrideB, colStrideB, beta, c, rowStrideC, cint(panelsPerDev), cint(ord(gather)), cint(flags))
  else:
    check laser_hip_gemm_strided_i64_sharded_dev(n, dv, M, N, K, alpha, a, rowStrideA, colStrideA, b, rowStrideB, colStrideB, beta, c, rowStrideC, cint(panelsPerDev), cint(ord(gather)), cint(flags))
proc allocPinned*[T](len: int): ptr UncheckedArray[T] =
  ## page-locked host memory for operands of the host-pointer calls (what an allocator would hand to Tensor[T])
  var p: pointer
  check laser_hip_host_alloc(p.addr, sizeof(T) * len)
  cast[ptr UncheckedArray[T]](p)
proc freePinned*(p: pointer) = check laser_hip_host_free(p)
proc registerPinned*(p: pointer, bytes: int) = check laser_hip_host_register(p, bytes)
proc unregisterPinned*(p: pointer) = check laser_hip_host_unregister(p)

# gemm_strided on device-resident operands: the SAME parameter list as gemm.nim:184-193 with DevicePtr[T]
# in place of ptr T (overload resolution keeps host and device pointers apart), asynchronous on `stream`.
proc gemm_strided*[T: SomeNumber](
      M, N, K: int,
      alpha: T,
      A: DevicePtr[T],
      rowStrideA, colStrideA: int,
      B: DevicePtr[T],
      rowStrideB, colStrideB: int,
      beta: T,
      C: DevicePtr[T],
      rowStrideC, colStrideC: int,
      stream: pointer = nil) =
  when T is float32:
    check laser_hip_gemm_strided_f32_dev(M, N, K, alpha, pointer(A), rowStrideA, colStrideA, pointer(B), rowStrideB, colStrideB, beta, pointer(C), rowStrideC, colStrideC, stream)
  elif T is float64:
    check laser_hip_gemm_strided_f64_dev(M, N, K, alpha, pointer(A), rowStrideA, colStrideA, pointer(B), rowStrideB, colStrideB, beta, pointer(C), rowStrideC, colStrideC, stream)
  elif T is int32:
    check laser_hip_gemm_strided_i32_dev(M, N, K, alpha, pointer(A), rowStrideA, colStrideA, pointer(B), rowStrideB, colStrideB, beta, pointer(C), rowStrideC, colStrideC, stream)
  elif T is uint32:
    check laser_hip_gemm_strided_i32_dev(M, N, K, cast[int32](alpha), pointer(A), rowStrideA, colStrideA, pointer(B), rowStrideB, colStrideB, cast[int32](beta), pointer(C), rowStrideC, colStrideC, stream)
  elif T is int64 or T is int:
    check laser_hip_gemm_strided_i64_dev(M, N, K, cast[int64](alpha), pointer(A), rowStrideA, colStrideA, pointer(B), rowStrideB, colStrideB, cast[int64](beta), pointer(C), rowStrideC, colStrideC, stream)
  else:
    {.error: "laser_hip: unsupported element type " & $T.}
