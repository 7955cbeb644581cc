"""ctypes binding of libtdsa_hip.so (include/tdsa_hip.h).

The HIP library IS the product path: if it is missing or fails to load this module raises
ImportError - there is deliberately no numpy fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDSA_HIP_LIB", os.path.join(_HERE, "libtdsa_hip.so"))

TDSA_OK = 0
IN_I8, IN_U8, IN_C64 = 0, 1, 2
DB_MAG, DB_POW = 0, 1
AVG_OFF, AVG_EXP, AVG_LIN = 0, 1, 2
HOLD_MAX, HOLD_MIN = 1, 2
CH_MONO, CH_LEFT, CH_RIGHT, CH_STEREO = 0, 1, 2, 3
RESET_AVG, RESET_HOLD_MAX, RESET_HOLD_MIN, RESET_DC, RESET_TARE, RESET_ALL = 1, 2, 4, 8, 16, 31


class Mode(C.Structure):
    _fields_ = [("db_mode", C.c_int32), ("power_scale", C.c_float), ("log_floor", C.c_float),
                ("avg_mode", C.c_int32), ("avg_n", C.c_int32), ("dc_alpha", C.c_float),
                ("cal_offset_db", C.c_float), ("hold_flags", C.c_uint32)]


class Info(C.Structure):
    _fields_ = [("nfft", C.c_int32), ("max_frames", C.c_int32), ("device_id", C.c_int32),
                ("grid", C.c_int32), ("block", C.c_int32), ("frames_per_block", C.c_int32),
                ("lds_bytes", C.c_int32), ("num_cu", C.c_int32),
                ("frames_held_max", C.c_int64), ("frames_held_min", C.c_int64),
                ("avg_count", C.c_int32), ("version", C.c_int32)]


# every symbol include/tdsa_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "tdsa_last_error_string": (C.c_char_p, []),
    "tdsa_version": (C.c_int, []),
    "tdsa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "tdsa_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tdsa_destroy": (C.c_int, [_P]),
    "tdsa_get_info": (C.c_int, [_P, C.POINTER(Info)]),
    "tdsa_set_window": (C.c_int, [_P, _P, C.c_int]),
    "tdsa_set_mode": (C.c_int, [_P, C.POINTER(Mode)]),
    "tdsa_reset_state": (C.c_int, [_P, C.c_uint32]),
    "tdsa_set_tare_baseline": (C.c_int, [_P, _P, C.c_int]),
    "tdsa_process_i8": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "tdsa_process_u8": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "tdsa_process_c64": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "tdsa_process_dev": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "tdsa_process_dev_batch": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_int, _P,
                                         C.c_size_t]),
    "tdsa_process_real2": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, C.c_int, _P]),
    "tdsa_real_input_supported": (C.c_int, [C.c_int]),
    "tdsa_get_hold": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int64)]),
    "tdsa_get_avg": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    "tdsa_host_register": (C.c_int, [_P, C.c_size_t]),
    "tdsa_host_unregister": (C.c_int, [_P]),
    "tdsa_welch_export": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "tdsa_welch_combine": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_int32), C.c_int, C.c_int, _P, _P]),
    "tdsa_peer_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_P), _P]),
    "tdsa_peer_free": (C.c_int, [C.c_int, _P]),
    "tdsa_peer_open": (C.c_int, [C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "tdsa_peer_can_access": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "tdsa_peer_close": (C.c_int, [C.c_int, _P]),
    "tdsa_welch_export_dev": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "tdsa_welch_combine_dev": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int, C.c_int, _P, _P]),
    "tdsa_shader_clock": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tdsa_get_dc": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tdsa_set_dc": (C.c_int, [_P, C.c_float, C.c_float]),
    "tdsa_synchronize": (C.c_int, [_P]),
    "tdsa_set_overlap": (C.c_int, [_P, C.c_int]),
    "tdsa_rows_stats": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P]),
    "tdsa_rows_top_peaks": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "tdsa_set_frame_stats": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "tdsa_get_frame_stats": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "tdsa_rows_marker_peaks": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                         _P, _P, _P, _P, _P]),
    "tdsa_density_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.POINTER(_P)]),
    "tdsa_density_destroy": (C.c_int, [_P]),
    "tdsa_density_set_decay": (C.c_int, [_P, C.c_float]),
    "tdsa_density_reset": (C.c_int, [_P]),
    "tdsa_density_update_dev": (C.c_int, [_P, _P, _P, C.c_int]),
    "tdsa_density_update": (C.c_int, [_P, _P, C.c_int]),
    "tdsa_density_read": (C.c_int, [_P, _P, C.c_int]),
    "tdsa_density_read_u8": (C.c_int, [_P, _P, _P]),
    "tdsa_waterfall_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(_P)]),
    "tdsa_waterfall_destroy": (C.c_int, [_P]),
    "tdsa_waterfall_push_dev": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "tdsa_waterfall_push": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "tdsa_waterfall_view": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    "tdsa_waterfall_view_u8": (C.c_int, [_P, C.c_float, C.c_float, _P]),
    "tdsa_pipe_create": (C.c_int, [_P, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(_P)]),
    "tdsa_pipe_destroy": (C.c_int, [_P]),
    "tdsa_pipe_acquire": (C.c_int, [_P, C.POINTER(_P)]),
    "tdsa_pipe_submit": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int]),
    "tdsa_pipe_collect": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]),
    "tdsa_pipe_collect_dev": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]),
    "tdsa_pipe_collect_u8": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int)]),
    "tdsa_pipe_set_levels": (C.c_int, [_P, C.c_float, C.c_float]),
    "tdsa_pipe_pending": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "tdsa_trace_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "tdsa_trace_destroy": (C.c_int, [_P]),
    "tdsa_trace_reset": (C.c_int, [_P, C.c_uint32]),
    "tdsa_trace_update": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                    _P, _P, _P, C.POINTER(C.c_int)]),
    "tdsa_trace_get_tare_baseline": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    "tdsa_trace_set_tare_baseline": (C.c_int, [_P, _P, C.c_int]),
    "tdsa_trace_avg_set_mode": (C.c_int, [_P, C.c_int, C.c_int]),
    "tdsa_trace_avg_process": (C.c_int, [_P, _P, C.c_int, _P, C.POINTER(C.c_int)]),
    "tdsa_dev_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_P)]),
    "tdsa_dev_free": (C.c_int, [C.c_int, _P]),
    "tdsa_memcpy_h2d": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "tdsa_memcpy_d2h": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "tdsa_profile_enable": (C.c_int, [_P, C.c_int]),
    "tdsa_profile_read": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "tdsa_timer_begin": (C.c_int, [_P]),
    "tdsa_timer_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "tdsa_debug_timeline": (C.c_int, [_P, _P]),
    "tdsa_debug_knob": (C.c_int, [_P, C.c_char_p, C.c_int]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the box
        raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class TdsaError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != TDSA_OK:
        raise TdsaError(f"tdsa error {rc}: {lib.tdsa_last_error_string().decode()}")
