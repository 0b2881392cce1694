"""ctypes binding of lib/pygmm.so (the C ABI in include/pygmm_hip.h).

Every pointer-returning function gets an explicit ``restype`` (the reference's wrapper does
not -- src/gmm/python/pygmm.py:30-31 -- and truncates handles to 32 bits on 64-bit Pythons).
There is no CPU fallback: if the library is missing, or no GPU is visible when a compute call
is made, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SR_PYGMM_LIB points at another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("SR_PYGMM_LIB") or os.path.join(_HERE, "lib", "pygmm.so")

LEGACY_SYMBOLS = ["new_gmm", "load", "dump", "train_model", "train_model_from_ubm", "score_all",
                  "score_batch", "score_instance", "get_dim", "get_nr_mixtures"]
EXT_SYMBOLS = [
    "sr_last_error", "sr_gpu_runtime_lost", "sr_device_count", "sr_set_device", "sr_set_thread_device", "sr_get_device", "sr_device_synchronize",
    "sr_device_name", "sr_device_numa_node", "sr_bind_thread_near_device", "sr_multi_slot_numa_node", "sr_free_gmm", "sr_gmm_from_arrays", "sr_gmm_get_params", "sr_gmm_dumps",
    "sr_gmm_loads", "sr_score_frames_f32", "sr_score_models_f32", "sr_modelset_create", "sr_modelset_free",
    "sr_modelset_size", "sr_modelset_info", "sr_modelset_dim", "sr_batch_from_pcm", "sr_batch_from_pcm_f32",
    "sr_batch_from_features", "sr_batch_update_pcm", "sr_batch_reset_pcm", "sr_batch_reset_features", "sr_batch_free", "sr_batch_num_utterances", "sr_batch_num_rows",
    "sr_batch_dim", "sr_batch_offsets", "sr_batch_download", "sr_score_batch_set",
    "sr_mfcc_create", "sr_mfcc_set_lpc", "sr_mfcc_free", "sr_mfcc_frame_len", "sr_mfcc_frame_shift",
    "sr_mfcc_num_frames", "sr_mfcc_tables", "sr_mfcc_extract_batch", "sr_predict_pcm_batch",
    "sr_train_f32", "sr_profile_enable", "sr_profile_reset", "sr_profile_get", "sr_set_option",
    "sr_last_score_kernel", "sr_last_em_stats_engine", "sr_ltsd_num_windows", "sr_ltsd_noise_spectrum", "sr_ltsd_compute", "sr_stream_create", "sr_stream_submit", "sr_stream_collect", "sr_stream_free",
    "sr_multi_create", "sr_multi_free", "sr_multi_slots", "sr_multi_slot_device", "sr_multi_predict_pcm",
    "sr_hbm_copy_gbps", "sr_reference_rand_sample", "sr_flush_stats", "sr_host_register", "sr_host_unregister",
    "sr_mfma_peak_probe", "sr_mfma_streamed_probe", "sr_kmeans_fast_stats",
]

SR_CLAMP_COMPAT = 1
SR_SCORE_PRECISE = 0x200
SR_STREAM_GRAPH = 0x100
T_SCORE, T_MFCC, T_CMVN, T_FINALIZE, T_ESTEP, T_SCORE_REF = 0, 1, 2, 3, 4, 5


class Parameter(C.Structure):
    """struct Parameter, src/gmm/src/pygmm.hh:12-26 (== GMMParameter, pygmm.py:18-27)."""
    _fields_ = [("nr_instance", C.c_int), ("nr_dim", C.c_int), ("nr_mixture", C.c_int),
                ("min_covar", C.c_double), ("threshold", C.c_double), ("nr_iteration", C.c_int),
                ("init_with_kmeans", C.c_int), ("concurrency", C.c_int), ("verbosity", C.c_int)]


class SRError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and prototype the shared library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SRError("%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                      "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    dpp = C.POINTER(C.POINTER(C.c_double))
    sig = {
        # legacy (pygmm.hh:28-41)
        "new_gmm": (vp, [i32, i32]),
        "load": (vp, [C.c_char_p]),
        "dump": (None, [vp, C.c_char_p]),
        "train_model": (None, [vp, dpp, C.POINTER(Parameter)]),
        "train_model_from_ubm": (None, [vp, vp, dpp, C.POINTER(Parameter)]),
        "score_all": (dbl, [vp, dpp, i32, i32, i32]),
        "score_batch": (None, [vp, dpp, dp, i32, i32, i32]),
        "score_instance": (dbl, [vp, dp, i32]),
        "get_dim": (i32, [vp]),
        "get_nr_mixtures": (i32, [vp]),
        # extensions
        "sr_last_error": (C.c_char_p, []),
        "sr_gpu_runtime_lost": (i32, []),
        "sr_device_count": (i32, []),
        "sr_set_device": (i32, [i32]),
        "sr_set_thread_device": (i32, [i32]),
        "sr_get_device": (i32, []),
        "sr_device_synchronize": (i32, []),
        "sr_device_name": (i32, [C.c_char_p, i32]),
        "sr_device_numa_node": (i32, [i32]),
        "sr_bind_thread_near_device": (i32, [i32]),
        "sr_multi_slot_numa_node": (i32, [vp, i32]),
        "sr_free_gmm": (None, [vp]),
        "sr_gmm_from_arrays": (vp, [i32, i32, dp, dp, dp]),
        "sr_gmm_get_params": (i32, [vp, dp, dp, dp]),
        "sr_gmm_dumps": (i32, [vp, C.c_char_p, C.c_long, C.POINTER(C.c_long)]),
        "sr_gmm_loads": (vp, [C.c_char_p]),
        "sr_score_frames_f32": (i32, [vp, fp, C.c_long, i32, fp, dp, i32]),
        "sr_score_models_f32": (i32, [C.POINTER(vp), i32, fp, C.c_long, i32, dp, i32]),
        "sr_modelset_create": (vp, [C.POINTER(vp), i32]),
        "sr_modelset_free": (None, [vp]),
        "sr_modelset_size": (i32, [vp]),
        "sr_modelset_info": (i32, [vp, dp]),
        "sr_modelset_dim": (i32, [vp]),
        "sr_batch_from_pcm": (vp, [C.POINTER(C.c_int16), C.POINTER(i64), i32]),
        "sr_batch_from_pcm_f32": (vp, [fp, C.POINTER(i64), i32]),
        "sr_batch_from_features": (vp, [fp, i64, i32, C.POINTER(i64), i32]),
        "sr_batch_update_pcm": (i32, [vp, C.POINTER(C.c_int16), i64]),
        "sr_batch_reset_pcm": (i32, [vp, C.POINTER(C.c_int16), C.POINTER(i64), i32]),
        "sr_batch_reset_features": (i32, [vp, fp, i64, i32, C.POINTER(i64), i32]),
        "sr_batch_free": (None, [vp]),
        "sr_batch_num_utterances": (i32, [vp]),
        "sr_batch_num_rows": (i64, [vp]),
        "sr_batch_dim": (i32, [vp]),
        "sr_batch_offsets": (i32, [vp, C.POINTER(i64)]),
        "sr_batch_download": (i32, [vp, fp]),
        "sr_score_batch_set": (i32, [vp, vp, dp, C.POINTER(i32), fp, i32]),
        "sr_mfcc_create": (vp, [dbl, dbl, dbl, i32, i32, i32, dbl]),
        "sr_mfcc_set_lpc": (i32, [vp, i32]),
        "sr_mfcc_free": (None, [vp]),
        "sr_mfcc_frame_len": (i32, [vp]),
        "sr_mfcc_frame_shift": (i32, [vp]),
        "sr_mfcc_num_frames": (i64, [vp, i64]),
        "sr_mfcc_tables": (i32, [vp, dp, dp, dp]),
        "sr_mfcc_extract_batch": (vp, [vp, vp, i32, i32]),
        "sr_predict_pcm_batch": (i32, [vp, vp, vp, i32, dp, C.POINTER(i32), i32]),
        "sr_train_f32": (i32, [vp, vp, fp, C.c_long, i32, C.POINTER(Parameter), C.c_long]),
        "sr_profile_enable": (i32, [i32]),
        "sr_profile_reset": (i32, []),
        "sr_profile_get": (i32, [i32, dp, C.POINTER(C.c_long)]),
        "sr_set_option": (i32, [C.c_char_p, C.c_long]),
        "sr_last_score_kernel": (C.c_char_p, []),
        "sr_last_em_stats_engine": (C.c_int, []),
        "sr_flush_stats": (None, [C.POINTER(C.c_long)] * 3),
        "sr_mfma_peak_probe": (i32, [C.c_double, dp, dp]),
        "sr_mfma_streamed_probe": (i32, [C.c_double, dp, dp]),
        "sr_kmeans_fast_stats": (None, [C.POINTER(C.c_long)] * 2),
        "sr_host_register": (i32, [vp, C.c_size_t]),
        "sr_host_unregister": (i32, [vp]),
        "sr_reference_rand_sample": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
        "sr_ltsd_num_windows": (i64, [i64, i32]),
        "sr_ltsd_noise_spectrum": (i32, [vp, i32, fp]),
        "sr_ltsd_compute": (i32, [vp, i32, i32, fp, fp, C.POINTER(i64)]),
        "sr_stream_create": (vp, [vp, vp, i32, i64, i32, i32]),
        "sr_stream_submit": (i32, [vp, C.POINTER(C.c_int16)]),
        "sr_stream_collect": (i32, [vp, dp, C.POINTER(i32), dp]),
        "sr_stream_free": (None, [vp]),
        "sr_multi_create": (vp, [C.POINTER(vp), i32, dbl, dbl, dbl, i32, i32, i32, dbl, i32]),
        "sr_multi_free": (None, [vp]),
        "sr_multi_slots": (i32, [vp]),
        "sr_multi_slot_device": (i32, [vp, i32]),
        "sr_multi_predict_pcm": (i32, [vp, C.POINTER(C.c_int16), C.POINTER(i64), i32, i32, dp, C.POINTER(i32), dp, i32]),
        "sr_hbm_copy_gbps": (i32, [C.c_size_t, i32, dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def gpu_runtime_lost() -> bool:
    """True in a process forked after its parent initialised the GPU runtime (include/pygmm_hip.h, "Processes"): the per-model
    entry points are served by a helper process there, the batched ones refuse."""
    return bool(lib().sr_gpu_runtime_lost())


def last_error() -> str:
    return lib().sr_last_error().decode("utf-8", "replace")


def check(status, what: str = "call"):
    """Raise on a negative status / NULL handle."""
    if status is None or (isinstance(status, int) and status < 0):
        raise SRError("%s failed: %s" % (what, last_error()))
    return status


# ---- small marshalling helpers ----

def f32_matrix(X) -> np.ndarray:
    a = np.ascontiguousarray(X, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("expected a 2-D [frames, dim] array, got shape %r" % (a.shape,))
    return a


def as_fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def as_dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def as_i64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def as_i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def set_device(device: int) -> None:
    check(lib().sr_set_device(int(device)), "sr_set_device")


def set_thread_device(device: int) -> None:
    """The calling host thread's current device (handles belong to the device they were created on)."""
    check(lib().sr_set_thread_device(int(device)), "sr_set_thread_device")


def hbm_copy_gbps(nbytes: int = 1 << 30, iters: int = 10) -> float:
    """Measured device-to-device copy rate (read + written bytes per second, GB/s)."""
    g = C.c_double(0)
    check(lib().sr_hbm_copy_gbps(int(nbytes), int(iters), C.byref(g)), "sr_hbm_copy_gbps")
    return g.value


def device_count() -> int:
    return int(lib().sr_device_count())


def bind_thread_near_device(device: int) -> int:
    """Pin the calling host thread to the cores of `device`'s NUMA node (sysfs); -> the node, -1 when left alone."""
    return int(lib().sr_bind_thread_near_device(int(device)))


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().sr_device_name(buf, 256), "sr_device_name")
    return buf.value.decode()


def synchronize() -> None:
    check(lib().sr_device_synchronize(), "sr_device_synchronize")


def set_option(key: str, value: int) -> None:
    check(lib().sr_set_option(key.encode(), int(value)), "sr_set_option")


def host_register(a) -> None:
    """Page-lock a numpy array's memory so that the copy engines read it in place (sr_multi_predict_pcm)."""
    check(lib().sr_host_register(C.c_void_p(a.ctypes.data), a.nbytes), "sr_host_register")


def host_unregister(a) -> None:
    check(lib().sr_host_unregister(C.c_void_p(a.ctypes.data)), "sr_host_unregister")


def flush_stats():
    """(resolve calls, (tile, model) pairs noted, frames re-evaluated) of the partial-product path (csrc/gmm_flush.hip)."""
    v = [C.c_long(0) for _ in range(3)]
    lib().sr_flush_stats(*[C.byref(x) for x in v])
    return tuple(int(x.value) for x in v)


def kmeans_fast_stats():
    """(full nearest-centre searches of the k-means initialiser taken the fast way, points those left to the exact pass)"""
    v = [C.c_long(0) for _ in range(2)]
    lib().sr_kmeans_fast_stats(*[C.byref(x) for x in v])
    return tuple(int(x.value) for x in v)


def mfma_peak_probe(ms_target: float = 50.0):
    """(executed fp16 MFMA TFLOP/s, shader clock in MHz) of a kernel that only issues v_mfma_f32_32x32x16_f16, on the current
    device, for about ms_target milliseconds: what the matrix pipe sustains under the socket's power cap (csrc/probe.hip)."""
    t, f = C.c_double(0.0), C.c_double(0.0)
    check(lib().sr_mfma_peak_probe(C.c_double(ms_target), C.byref(t), C.byref(f)), "sr_mfma_peak_probe")
    return float(t.value), float(f.value)


def mfma_streamed_probe(ms_target: float = 50.0):
    """The same, with the chains fed as the scoring kernel feeds them: a fresh A fragment from LDS for every MFMA, random operand
    bits (csrc/probe.hip, mode 1)."""
    t, f = C.c_double(0.0), C.c_double(0.0)
    check(lib().sr_mfma_streamed_probe(C.c_double(ms_target), C.byref(t), C.byref(f)), "sr_mfma_streamed_probe")
    return float(t.value), float(f.value)


def last_score_kernel() -> str:
    return lib().sr_last_score_kernel().decode()


def last_em_stats_engine() -> int:
    """1 vector ALU, 2 fp64 matrix cores, 3 the same with the responsibilities on the 16-bit matrix cores (0: no E-step yet)"""
    return int(lib().sr_last_em_stats_engine())


def profile_enable(on: bool = True) -> None:
    lib().sr_profile_enable(1 if on else 0)


def profile_reset() -> None:
    check(lib().sr_profile_reset(), "sr_profile_reset")


def profile_get(kind: int):
    ms = C.c_double(0)
    n = C.c_long(0)
    check(lib().sr_profile_get(kind, C.byref(ms), C.byref(n)), "sr_profile_get")
    return ms.value, n.value
