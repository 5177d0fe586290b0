"""ctypes loader for liblaser_hip.so (the C-ABI in include/laser_hip.h).

There is NO fallback: if the shared library is missing or the GPU is not a gfx950 the calls raise.
"""
import ctypes as C
import os

ABI_VERSION = 2   # include/laser_hip.h LASER_HIP_ABI_VERSION this mirror was written against

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblaser_hip.so")

OK, E_INVALID, E_HIP, E_NODEVICE, E_HANDLE = 0, 1, 2, 3, 4
F32_LASER_ORDER, F32_FAST = 0, 1

_CT = {"f32": C.c_float, "f64": C.c_double, "i32": C.c_int32, "i64": C.c_int64}


class LaserHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"laser_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load the library (once) and declare every prototype of include/laser_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C laser_amd/csrc`. laser_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    i64, vp, ci = C.c_int64, C.c_void_p, C.c_int
    L.laser_hip_init.argtypes = [ci]
    L.laser_hip_last_error.restype = C.c_char_p
    L.laser_hip_version.restype = C.c_char_p
    if L.laser_hip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"liblaser_hip.so has ABI {L.laser_hip_abi_version()}, this mirror was written against {ABI_VERSION} (rebuild: make -C laser_amd/csrc)")
    L.laser_hip_arch.restype = C.c_char_p
    L.laser_hip_plan_f32.argtypes = [i64, i64, i64, ci, ci, C.POINTER(i64)]
    L.laser_hip_set_float_mode.argtypes = [ci]
    L.laser_hip_set_f32_config.argtypes = [ci]
    L.laser_hip_set_option.argtypes = [C.c_char_p, ci]
    L.laser_hip_get_option.argtypes = [C.c_char_p, C.POINTER(i64)]
    L.laser_hip_f32_config_name.argtypes = [ci]
    L.laser_hip_f32_config_name.restype = C.c_char_p
    for sfx, ct in _CT.items():
        g = [i64, i64, i64, ct, vp, i64, i64, vp, i64, i64, ct, vp, i64, i64]
        getattr(L, f"laser_hip_gemm_strided_{sfx}").argtypes = g
        getattr(L, f"laser_hip_gemm_strided_{sfx}_dev").argtypes = g + [vp]
        getattr(L, f"laser_hip_gemm_strided_batched_{sfx}_dev").argtypes = [
            i64, i64, i64, i64, ct, vp, i64, i64, i64, vp, i64, i64, i64, ct, vp, i64, i64, i64, vp]
        for ab in "AB":
            f = getattr(L, f"laser_hip_gemm_prepack{ab}_mem_required_{sfx}")
            f.argtypes = [i64, i64, i64]
            f.restype = i64
            getattr(L, f"laser_hip_gemm_prepack{ab}_{sfx}").argtypes = [vp, i64, i64, i64, vp, i64, i64]
            getattr(L, f"laser_hip_gemm_prepack{ab}_{sfx}_dev").argtypes = [vp, i64, i64, i64, vp, i64, i64, vp]
        getattr(L, f"laser_hip_gemm_packed_{sfx}").argtypes = [i64, i64, i64, ct, vp, vp, ct, vp, i64, i64]
        getattr(L, f"laser_hip_gemm_packed_{sfx}_dev").argtypes = [i64, i64, i64, ct, vp, vp, ct, vp, i64, i64, vp]
        pp = C.POINTER(vp)  # table of per-device pointers
        getattr(L, f"laser_hip_gemm_strided_{sfx}_sharded").argtypes = [ci, C.POINTER(ci)] + g
        getattr(L, f"laser_hip_gemm_strided_{sfx}_sharded_dev").argtypes = [
            ci, C.POINTER(ci), i64, i64, i64, ct, pp, i64, i64, pp, i64, i64, ct, pp, i64, ci, ci, ci]
    for sfx, ct in _CT.items():
        pi = C.POINTER(i64)
        getattr(L, f"laser_hip_map_strided_unary_{sfx}_dev").argtypes = [ci, vp, pi, vp, pi, pi, ci, ct, ct, vp]
        getattr(L, f"laser_hip_map_strided_binary_{sfx}_dev").argtypes = [ci, vp, pi, vp, pi, vp, pi, pi, ci, ct, ct, vp]
    L.laser_hip_host_alloc.argtypes = [C.POINTER(vp), i64]
    L.laser_hip_host_free.argtypes = [vp]
    L.laser_hip_host_register.argtypes = [vp, i64]
    L.laser_hip_host_unregister.argtypes = [vp]
    L.laser_hip_shard_plan.argtypes = [i64, ci, ci, C.POINTER(i64), C.POINTER(ci), C.POINTER(i64)]
    L.laser_hip_set_shard_devices.argtypes = [ci]
    for sfx in ("f32", "f64"):  # fused epilogue: + bias view (ptr, rowStride, colStride) + activation
        ct = _CT[sfx]
        g = [i64, i64, i64, ct, vp, i64, i64, vp, i64, i64, ct, vp, i64, i64, vp, i64, i64, ci]
        getattr(L, f"laser_hip_gemm_strided_ex_{sfx}").argtypes = g
        getattr(L, f"laser_hip_gemm_strided_ex_{sfx}_dev").argtypes = g + [vp]
    L.laser_hip_gemm_prepack_release.argtypes = [vp]
    for b in ("b32", "b64", "b16", "b8"):
        getattr(L, f"laser_hip_transpose2d_copy_{b}").argtypes = [vp, vp, i64, i64]
        getattr(L, f"laser_hip_transpose2d_batched_{b}").argtypes = [vp, vp, i64, i64, i64]
        getattr(L, f"laser_hip_nchw2nhwc_{b}").argtypes = [vp, vp, i64, i64, i64, i64]
        getattr(L, f"laser_hip_nhwc2nchw_{b}").argtypes = [vp, vp, i64, i64, i64, i64]
        getattr(L, f"laser_hip_transpose2d_batched_{b}_dev").argtypes = [vp, vp, i64, i64, i64, vp]
    L.laser_hip_conv2d_out_shape.argtypes = [i64] * 12 + [C.POINTER(i64)] * 4
    L.laser_hip_im2col_workspace_size.argtypes = [i64] * 12
    L.laser_hip_im2col_workspace_size.restype = i64
    L.laser_hip_im2col_f32.argtypes = [vp, i64, i64, vp] + [i64] * 9
    L.laser_hip_im2col_f32_dev.argtypes = [vp, i64, i64, vp] + [i64] * 10 + [vp]
    L.laser_hip_im2col_f64.argtypes = [vp, i64, i64, vp] + [i64] * 9
    L.laser_hip_im2col_f64_dev.argtypes = [vp, i64, i64, vp] + [i64] * 10 + [vp]
    L.laser_hip_conv2d_im2col_f32.argtypes = [vp, vp, i64, i64, i64, i64, vp] + [i64] * 8 + [vp]
    L.laser_hip_conv2d_im2col_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, vp] + [i64] * 8 + [vp, vp]
    L.laser_hip_conv2d_im2col_ex_f32.argtypes = [vp, vp, i64, i64, i64, i64, vp] + [i64] * 8 + [vp, vp, ci]
    L.laser_hip_conv2d_im2col_ex_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, vp] + [i64] * 8 + [vp, vp, ci, vp]
    L.laser_hip_storage_alloc.argtypes = [C.POINTER(vp), i64]
    L.laser_hip_storage_free.argtypes = [vp]
    L.laser_hip_storage_upload.argtypes = [vp, vp, i64]
    L.laser_hip_storage_download.argtypes = [vp, vp, i64]
    L.laser_hip_storage_set_zero.argtypes = [vp, i64, vp]
    L.laser_hip_storage_alloc_stream.argtypes = [C.POINTER(vp), i64, vp]
    L.laser_hip_storage_upload_stream.argtypes = [vp, vp, i64, vp]
    L.laser_hip_storage_download_stream.argtypes = [vp, vp, i64, vp]
    for b in ("b32", "b64"):
        getattr(L, f"laser_hip_copy_strided_{b}_dev").argtypes = [vp, C.POINTER(i64), vp, C.POINTER(i64), C.POINTER(i64), ci, vp]
    L.laser_hip_cblas_sgemm.argtypes = [ci, ci, ci, i64, i64, i64, C.c_float, vp, i64, vp, i64, C.c_float, vp, i64]
    L.laser_hip_cblas_dgemm.argtypes = [ci, ci, ci, i64, i64, i64, C.c_double, vp, i64, vp, i64, C.c_double, vp, i64]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise LaserHipError(rc, lib().laser_hip_last_error().decode(errors="replace"))


def ctype_of(sfx):
    return _CT[sfx]


# Every symbol include/laser_hip.h declares (tests check the .so exports each of them).
def declared_symbols():
    names = ["laser_hip_init", "laser_hip_finalize", "laser_hip_last_error", "laser_hip_version", "laser_hip_abi_version",
             "laser_hip_device_count", "laser_hip_plan_f32", "laser_hip_arch", "laser_hip_set_float_mode",
             "laser_hip_get_float_mode", "laser_hip_set_f32_config", "laser_hip_f32_config_count",
             "laser_hip_f32_config_name", "laser_hip_set_option", "laser_hip_get_option", "laser_hip_gemm_prepack_release",
             "laser_hip_conv2d_out_shape", "laser_hip_im2col_workspace_size", "laser_hip_im2col_f32",
             "laser_hip_im2col_f32_dev", "laser_hip_conv2d_im2col_f32", "laser_hip_conv2d_im2col_f32_dev",
             "laser_hip_cblas_sgemm", "laser_hip_cblas_dgemm",
             "laser_hip_gemm_strided_ex_f32", "laser_hip_gemm_strided_ex_f32_dev",
             "laser_hip_gemm_strided_ex_f64", "laser_hip_gemm_strided_ex_f64_dev",
             "laser_hip_conv2d_im2col_ex_f32", "laser_hip_conv2d_im2col_ex_f32_dev",
             "laser_hip_storage_alloc", "laser_hip_storage_free", "laser_hip_storage_trim", "laser_hip_storage_upload",
             "laser_hip_storage_download", "laser_hip_storage_set_zero",
             "laser_hip_storage_alloc_stream", "laser_hip_storage_upload_stream", "laser_hip_storage_download_stream",
             "laser_hip_copy_strided_b32_dev", "laser_hip_copy_strided_b64_dev",
             "laser_hip_host_alloc", "laser_hip_host_free", "laser_hip_host_register", "laser_hip_host_unregister",
             "laser_hip_shard_plan", "laser_hip_set_shard_devices", "laser_hip_get_shard_devices"]
    for s in _CT:
        names += [f"laser_hip_gemm_strided_{s}", f"laser_hip_gemm_strided_{s}_dev",
                  f"laser_hip_gemm_strided_batched_{s}_dev", f"laser_hip_gemm_packed_{s}",
                  f"laser_hip_gemm_strided_{s}_sharded", f"laser_hip_gemm_strided_{s}_sharded_dev",
                  f"laser_hip_map_strided_unary_{s}_dev", f"laser_hip_map_strided_binary_{s}_dev",
                  f"laser_hip_gemm_packed_{s}_dev"]
        for ab in "AB":
            names += [f"laser_hip_gemm_prepack{ab}_mem_required_{s}", f"laser_hip_gemm_prepack{ab}_{s}",
                      f"laser_hip_gemm_prepack{ab}_{s}_dev"]
    names += ["laser_hip_im2col_f64", "laser_hip_im2col_f64_dev"]
    for b in ("b32", "b64", "b16", "b8"):
        names += [f"laser_hip_transpose2d_copy_{b}", f"laser_hip_transpose2d_batched_{b}",
                  f"laser_hip_nchw2nhwc_{b}", f"laser_hip_nhwc2nchw_{b}",
                  f"laser_hip_transpose2d_batched_{b}_dev"]
    return names
