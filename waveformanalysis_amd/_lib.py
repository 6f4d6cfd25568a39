"""ctypes binding of libwfa_hip.so (C ABI: include/wfa_hip.h).

There is no CPU fallback: if the shared library is missing or a symbol is absent this module
raises, and so does everything that needs the device.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwfa_hip.so")

WFA_OK = 0
WFA_E_INVALID = -1
WFA_E_HIP = -2
WFA_E_STATE = -3
WFA_E_NOMEM = -4
WFA_E_RCCL = -5
WFA_E_LIMIT = -6

SRC_RAW, SRC_F32, SRC_SG_FUSED = 0, 1, 2
POL_UNKNOWN, POL_NEGATIVE, POL_POSITIVE, POL_POSITIVE_WAVE = 0, 1, 2, 3
ABI_VERSION = 1

_p = C.c_void_p
_i32, _i64, _f64, _int = C.c_int32, C.c_int64, C.c_double, C.c_int

# name -> (restype, argtypes); every symbol include/wfa_hip.h declares
SIGNATURES = {
    "wfa_abi_version": (_int, []),
    "wfa_device_count": (_int, [C.POINTER(_int)]),
    "wfa_last_error": (_int, [C.c_char_p, C.c_size_t]),
    "wfa_ctx_create": (_int, [_int, C.POINTER(_p)]),
    "wfa_ctx_destroy": (None, [_p]),
    "wfa_sync": (_int, [_p]),
    "wfa_release_scratch": (_int, [_p, C.POINTER(_i64)]),
    "wfa_set_option": (_int, [_p, C.c_char_p, _int]),
    "wfa_last_h2d_rate": (_int, [_p, C.POINTER(_f64)]),
    "wfa_hit_rows_source": (_int, [_p, _int]),
    "wfa_upload_pool_u16": (_int, [_p, _p, _i64]),
    "wfa_upload_pool_f32": (_int, [_p, _p, _i64]),
    "wfa_upload_records_soa": (_int, [_p, _i64] + [_p] * 10),
    "wfa_upload_records_packed": (_int, [_p, _p, _i64, _i32, _p, _i32, _f64, _p, _p, C.POINTER(_i32), C.POINTER(_int)]),
    "wfa_set_sg_plan": (_int, [_p, _int, _int, _p, _p, _int, _p, _i32, _i32, _i64, _i64]),
    "wfa_baseline_mean": (_int, [_p, _i32, _i32, _int, _p]),
    "wfa_filter_keep_output": (_int, [_p, _int]),
    "wfa_download_pool_f32": (_int, [_p, _p, _i64]),
    "wfa_savgol": (_int, [_p, _p]),
    "wfa_sosfiltfilt": (_int, [_p, _int, _p, _p, _i32, _p]),
    "wfa_threshold_hits_count": (_int, [_p, _int, _i32, _i32, _i32, C.POINTER(_i64)]),
    "wfa_threshold_hits_fill": (_int, [_p, _p, _i64]),
    "wfa_fused_baseline_filter_hits": (_int, [_p, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i64)]),
    "wfa_hits_enqueue": (_int, [_p, _int, _i32, _i32, _i32, _i32, _i32]),
    "wfa_hits_wait": (_int, [_p, C.POINTER(_i64)]),
    "wfa_find_peaks_count": (_int, [_p, _int, _int, _int, _f64, _int, _f64, _i32, _f64, _f64, _int, _i32, C.POINTER(_i64)]),
    "wfa_find_peaks_fill": (_int, [_p, _p, _i64]),
    "wfa_csv_decode_count": (_int, [_p, _p, _i64, _int, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "wfa_csv_decode_fill": (_int, [_p, _i64, _i32, _p, _p, _p, _p, _p, _p, _i64]),
    "wfa_find_hits_count": (_int, [_p, _int, _i64, _i32, _p, _f64, C.POINTER(_i64)]),
    "wfa_find_hits_fill": (_int, [_p, _i64, _p, _p]),
    "wfa_waveform_width": (_int, [_p, _int, _i64, _p, _p, _i64, _i32, _f64, _f64, _f64, _f64, _f64, _int, _p, _p]),
    "wfa_hit_merge_count": (_int, [_p, _i64] + [_p] * 7 + [_f64, _f64, C.POINTER(_i64)]),
    "wfa_hit_merge_fill": (_int, [_p, _i64, _i64, _p, _p]),
    "wfa_hit_merge_emit": (_int, [_p, _i64] + [_p] * 6 + [_i64, _p, _i64, _p] + [_p] * 6),
    "wfa_group_hit_windows_count": (_int, [_p, _i64] + [_p] * 10 + [_f64, C.POINTER(_i64)]),
    "wfa_group_hit_windows_fill": (_int, [_p, _i64, _i64] + [_p] * 4),
    "wfa_group_multi_channel_count": (_int, [_p, _i64, _p, _p, _f64, C.POINTER(_i64)]),
    "wfa_group_multi_channel_fill": (_int, [_p, _i64, _i64, _p, _p]),
    "wfa_v1725_index": (_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, C.POINTER(_i64)]),
    "wfa_records_sort": (_int, [_p, _i64, _p, _p, _p, _p, _p]),
    "wfa_pool_gather": (_int, [_p, _i64, _p, _p, _p, _i64, _p, _p, _i64]),
    "wfa_basic_features": (_int, [_p, _int, _i64, _i64, _int, _i64, _i64, _int, _p, _p]),
    "wfa_width_integral": (_int, [_p, _int, _f64, _f64, _f64, _p]),
    "wfa_features_both": (_int, [_p, _i64, _i64, _int, _i64, _i64, _int, _f64, _f64, _f64, _p, _p]),
    "wfa_profile_enable": (_int, [_p, _int]),
    "wfa_profile_reset": (_int, [_p]),
    "wfa_profile_get": (_int, [_p, _int, C.c_char_p, C.c_size_t, C.POINTER(_f64), C.POINTER(_i64)]),
    "wfa_rccl_unique_id": (_int, [_p]),
    "wfa_rccl_init": (_int, [_p, _int, _int, _p]),
    "wfa_rccl_allgather_counts": (_int, [_p, _i64, _p]),
    "wfa_rccl_gather_rows": (_int, [_p, _p, _i64, _i32, _int, _p, _p]),
    "wfa_rccl_gather_append": (_int, [_p, _int]),
    "wfa_rccl_destroy": (_int, [_p]),
}

_lib = None


class WfaError(RuntimeError):
    """HIP / RCCL / state failure reported by libwfa_hip.so."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libwfa_hip error {code}: {message}")
        self.code = code


def load() -> C.CDLL:
    """Load libwfa_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)  # RTLD_LOCAL: torch ships its own HIP/RCCL copies, keep the symbol spaces apart
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.wfa_abi_version() != ABI_VERSION:
        raise ImportError(f"libwfa_hip.so ABI {lib.wfa_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().wfa_last_error(buf, len(buf))
    return buf.value.decode(errors="replace")


def check(rc: int) -> None:
    """Map a return code to the exception the reference would raise for the same condition."""
    if rc == WFA_OK:
        return
    msg = last_error()
    if rc in (WFA_E_INVALID, WFA_E_LIMIT):
        raise ValueError(msg)
    raise WfaError(rc, msg)
