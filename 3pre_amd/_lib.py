"""ctypes binding of lib/libpre3.so (the C ABI of include/pre3.h).

There is no CPU fallback: if the shared library is missing this module raises at import, and
every compute call raises Pre3Error when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

try:  # torch ships its own libamdhip64.so.7; load it first so both share ONE HIP runtime
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the library
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRE3_LIB") or os.path.join(_HERE, "lib", "libpre3.so")      # PRE3_LIB: A/B builds while tuning

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "3pre_amd: %s is missing -- build it with `make -C 3pre_amd/csrc` (or __graft_entry__.build()); "
        "there is no CPU fallback for the HIP path" % LIB_PATH)

lib = C.CDLL(LIB_PATH)
lib.pre3_last_error.restype = C.c_char_p
lib.pre3_version.restype = C.c_char_p
lib.pre3_match_bench_create.restype = C.c_void_p
lib.pre3_match_bench_create_cls.restype = C.c_void_p
lib.pre3_hypothesis_support.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
lib.pre3_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_void_p]
lib.pre3_step_predicted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_void_p]

F64, F32 = 0, 1
INVDEPTH, CARTESIAN = 0, 1
X_K_K, X_K_KM1 = 0, 1


class Cam(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("f", "Cx", "Cy", "k1", "k2", "nRows", "nCols")]


class Pre3Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpre3 error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != 0:
        raise Pre3Error(rc, lib.pre3_last_error().decode(errors="replace"))


def dptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def addr(a):
    """Address of a C-contiguous array as an int: 0.3 us through the buffer protocol instead of 2 us through ndarray.ctypes
    (the per-step wrapper overhead is GPU idle time); falls back for read-only or empty arrays."""
    try:
        return C.addressof(C.c_char.from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def device_count():
    return int(lib.pre3_device_count())


# every symbol include/pre3.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "pre3_last_error", "pre3_device_count", "pre3_version", "pre3_create", "pre3_destroy", "pre3_sync", "pre3_set_cam",
    "pre3_set_map", "pre3_state_size", "pre3_set_state", "pre3_get_state", "pre3_predict", "pre3_predict_dense", "pre3_project", "pre3_innovation",
    "pre3_get_landmark_fields", "pre3_window_gate", "pre3_set_measurements", "pre3_ransac", "pre3_hypothesis_support", "pre3_ransac_score",
    "pre3_ransac_select", "pre3_ransac_export", "pre3_ransac_import", "pre3_update_li", "pre3_rescue", "pre3_update_hi", "pre3_update_all", "pre3_get_flags",
    "pre3_set_flags", "pre3_step", "pre3_step_all", "pre3_step_predicted", "pre3_set_option", "pre3_get_option", "pre3_map_delete", "pre3_map_add_inverse_depth", "pre3_map_inversedepth_2_cartesian", "pre3_map_management", "pre3_get_map", "pre3_set_descriptors", "pre3_get_descriptors", "pre3_set_scan", "pre3_ic_search", "pre3_vo_ransac", "pre3_vo_ransac_frames", "pre3_vo_bench", "pre3_update_ell", "pre3_siftmatch_f64", "pre3_siftmatch_f32", "pre3_siftmatch_u8",
    "pre3_siftmatch_i8", "pre3_siftmatch_partial", "pre3_siftmatch_merge", "pre3_match_shard_create", "pre3_match_shard_create_cls", "pre3_match_shard_run", "pre3_match_shard_merge",
    "pre3_match_shard_destroy", "pre3_release_scratch", "pre3_knn_f64", "pre3_timer_start",
    "pre3_timer_stop", "pre3_kernel_timing", "pre3_kernel_timing_read", "pre3_kernel_timing_info", "pre3_bench_downdate",
    "pre3_match_bench_create", "pre3_match_bench_create_cls", "pre3_match_bench_info", "pre3_match_bench_run", "pre3_match_bench_fetch", "pre3_match_bench_destroy",
    "pre3_comm_unique_id", "pre3_comm_create", "pre3_comm_destroy", "pre3_comm_info", "pre3_comm_set_timeout", "pre3_set_comm", "pre3_comm_init", "pre3_match_shard_set_comm",
    "pre3_ransac_sharded", "pre3_match_shard_match",
]
