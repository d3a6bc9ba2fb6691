"""ctypes binding of the C ABI in include/m3hip.h (libm3hip.so) and of the synthetic
generator (libm3synth.so).  Nothing here computes: every call lands in the HIP library,
and loading fails loudly when the library is missing (there is no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")

# symbols include/m3hip.h declares; tests check every one is exported
M3HIP_SYMBOLS = [
    "m3hip_create", "m3hip_destroy", "m3hip_last_error", "m3hip_set_stream", "m3hip_mem_info",
    "m3hip_alloc_pinned", "m3hip_free_pinned",
    "m3hip_monolith_begin", "m3hip_monolith_append", "m3hip_monolith_end", "m3hip_upload_monolith",
    "m3hip_eval_splines", "m3hip_sync",
    "m3hip_set_spline_param_map", "m3hip_set_binning", "m3hip_upload_events", "m3hip_set_osc_table",
    "m3hip_set_data", "m3hip_set_test_statistic",
    "m3hip_binned_upload", "m3hip_binned_upload_events", "m3hip_get_binned_weights",
    "m3hip_binned_upload_f64", "m3hip_get_binned_weights_f64", "m3hip_set_osc_table_f64", "m3hip_set_osc_table_device", "m3hip_set_osc_table_device_f64", "m3hip_get_event_weights_f64",
    "m3hip_set_binning_edges", "m3hip_upload_kinematics", "m3hip_compute_bins", "m3hip_add_kin_shift", "m3hip_add_kin_shift_aux",
    "m3hip_clear_kin_shifts", "m3hip_add_kin_cut", "m3hip_get_event_bins",
    "m3hip_binning_begin", "m3hip_binning_add_uniform", "m3hip_binning_add_nonuniform", "m3hip_binning_end",
    "m3hip_reweight", "m3hip_get_likelihood", "m3hip_step",
    "m3hip_get_bins", "m3hip_get_spline_weights", "m3hip_record_event_weights", "m3hip_get_event_weights",
    "m3hip_get_segments",
    "m3hip_comm_get_unique_id", "m3hip_comm_init", "m3hip_comm_destroy", "m3hip_comm_available", "m3hip_device_synchronize", "m3hip_upload_event_weights", "m3hip_upload_weight_index",
    "m3hip_set_weight_table", "m3hip_set_deterministic", "m3hip_monolith_expect_events", "m3hip_monolith_order_key",
    "m3hip_comm_set_validate", "m3hip_comm_last_validation", "m3hip_xch_create", "m3hip_xch_open", "m3hip_xch_destroy", "m3hip_xch_set_timeout", "m3hip_device_count", "m3hip_par_create", "m3hip_par_destroy", "m3hip_par_upload", "m3hip_par_launch", "m3hip_par_result",
    "m3hip_debug_layout", "m3hip_debug_short_brick_plan", "m3hip_debug_set_bins", "m3hip_debug_host_times", "m3hip_debug_set_knobs", "m3hip_debug_set_hist_replicas", "m3hip_debug_device_span",
    "m3hip_get_info", "m3hip_timing_enable", "m3hip_timing_read", "m3hip_timing_reset", "m3hip_version",
]

kPoisson, kBarlowBeeston, kIceCube, kPearson, kDembinskiAbdelmotteleb = range(5)
LARGE_LOGL = 1234567890.0
LOW_MC_BOUND = 0.00001
UNDER_OVERFLOW_BIN = -1
UNIQUE_ID_BYTES = 128
XCH_HANDLE_BYTES = 64


class M3HipError(RuntimeError):
    """Raised where the reference would throw MaCh3Exception (Manager/MaCh3Exception.h:23-49)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"m3hip error {code}: {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [
        ("n_events", C.c_uint64), ("n_splines", C.c_uint64), ("n_tf1", C.c_uint64),
        ("n_tiles_uniform", C.c_uint64), ("n_tiles_brick", C.c_uint64),
        ("coeff_bytes", C.c_uint64), ("alg_bytes_eval", C.c_uint64), ("alg_bytes_fused", C.c_uint64),
        ("n_params", C.c_int), ("max_knots", C.c_int), ("n_bins", C.c_int), ("n_samples", C.c_int),
        ("n_tiles_padded", C.c_uint64), ("pad_responses", C.c_uint64), ("n_grouped_chunks", C.c_uint64), ("brick_groups", C.c_uint64),
        ("cached_stream_bytes", C.c_uint64), ("tf1_codes_around", C.c_uint64), ("osc_lines_per_kevent", C.c_double),
        ("n_keyed_chunks", C.c_uint64), ("n_key_refused", C.c_uint64),
    ]


class SynthConfig(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("n_params", C.c_int32), ("n_knots", C.c_int32), ("mixed", C.c_int32), ("n_norm", C.c_int32),
        ("max_norm_per_event", C.c_int32), ("n_samples", C.c_int32), ("n_bins", C.c_int32), ("osc_n", C.c_int32),
        ("occupancy", C.c_double), ("bin_lo", C.c_double), ("bin_hi", C.c_double), ("nc_fraction", C.c_double),
        ("osc_sorted_events", C.c_uint64),
    ]


_hip = None
_synth = None


def lib_path(name: str) -> str:
    return os.path.join(_LIBDIR, name)


def _ptr(a, dtype):
    """numpy array -> typed pointer (None passes NULL); asserts dtype and contiguity."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def load_m3hip():
    """Load libm3hip.so (built by `make` / __graft_entry__.build()).  No fallback."""
    global _hip
    if _hip is not None:
        return _hip
    path = os.environ.get("M3HIP_LIB") or lib_path("libm3hip.so")  # M3HIP_LIB: A/B runs of two builds (scripts/ab_step.py)
    if not os.path.exists(path):
        raise M3HipError(-2, f"{path} is missing: build it with `make` (hipcc --offload-arch=gfx950); "
                             "this engine has no CPU fallback")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    sig = {
        "m3hip_create": [i32, C.POINTER(vp)],
        "m3hip_destroy": [vp],
        "m3hip_set_stream": [vp, vp],
        "m3hip_mem_info": [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
        "m3hip_alloc_pinned": [vp, C.POINTER(vp), C.c_size_t],
        "m3hip_free_pinned": [vp, vp],
        "m3hip_monolith_begin": [vp, i32, i32, vp, vp],
        "m3hip_monolith_append": [vp, u64, vp, u64, vp, vp, u64, vp, vp, u64, vp, vp],
        "m3hip_monolith_order_key": [vp, vp, u64],
        "m3hip_monolith_end": [vp],
        "m3hip_upload_monolith": [vp, u64, i32, i32, vp, vp, vp, u64, vp, vp, u64, vp, vp, u64, vp, vp],
        "m3hip_eval_splines": [vp, vp, vp, vp],
        "m3hip_sync": [vp],
        "m3hip_set_spline_param_map": [vp, vp, i32],
        "m3hip_set_binning": [vp, i32, vp],
        "m3hip_upload_events": [vp, u64, u64, vp, vp, vp, vp],
        "m3hip_set_osc_table": [vp, vp, u64],
        "m3hip_set_binning_edges": [vp, i32, vp, vp, vp],
        "m3hip_binning_begin": [vp],
        "m3hip_binning_add_uniform": [vp, i32, vp, vp],
        "m3hip_binning_add_nonuniform": [vp, i32, i32, vp],
        "m3hip_binning_end": [vp],
        "m3hip_binned_upload": [vp, u64, i32, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp],
        "m3hip_binned_upload_events": [vp, u64, u64, vp, vp],
        "m3hip_get_binned_weights": [vp, vp, i32],
        "m3hip_binned_upload_f64": [vp, u64, i32, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp],
        "m3hip_get_binned_weights_f64": [vp, vp, i32],
        "m3hip_set_osc_table_f64": [vp, vp, u64],
        "m3hip_set_osc_table_device": [vp, vp, u64],
        "m3hip_set_osc_table_device_f64": [vp, vp, u64],
        "m3hip_get_event_weights_f64": [vp, vp, u64, u64],
        "m3hip_upload_kinematics": [vp, u64, u64, vp, vp, i32],
        "m3hip_compute_bins": [vp],
        "m3hip_add_kin_shift": [vp, i32, i32, i32, i32],
        "m3hip_add_kin_shift_aux": [vp, i32, i32, i32, i32],
        "m3hip_clear_kin_shifts": [vp],
        "m3hip_add_kin_cut": [vp, i32, i32, C.c_double, C.c_double],
        "m3hip_get_event_bins": [vp, vp, u64, u64],
        "m3hip_set_data": [vp, vp, i32],
        "m3hip_set_test_statistic": [vp, i32, i32],
        "m3hip_reweight": [vp, vp, i32],
        "m3hip_get_likelihood": [vp, C.POINTER(C.c_double), vp],
        "m3hip_step": [vp, vp, i32, C.POINTER(C.c_double), vp],
        "m3hip_get_bins": [vp, vp, vp, i32],
        "m3hip_get_spline_weights": [vp, vp, u64, u64],
        "m3hip_record_event_weights": [vp, i32],
        "m3hip_get_event_weights": [vp, vp, u64, u64],
        "m3hip_get_segments": [vp, vp, vp, i32],
        "m3hip_comm_get_unique_id": [vp],
        "m3hip_comm_init": [vp, vp, i32, i32],
        "m3hip_comm_destroy": [vp],
        "m3hip_comm_available": [],
        "m3hip_device_synchronize": [],
        "m3hip_upload_event_weights": [vp, i32, u64, u64, vp],
        "m3hip_upload_weight_index": [vp, i32, u64, u64, vp],
        "m3hip_set_weight_table": [vp, i32, vp, u64],
        "m3hip_set_deterministic": [vp, i32],
        "m3hip_monolith_expect_events": [vp, u64],
        "m3hip_comm_set_validate": [vp, i32],
        "m3hip_comm_last_validation": [vp, vp, vp],
        "m3hip_xch_create": [vp, i32, i32, vp],
        "m3hip_xch_open": [vp, vp],
        "m3hip_xch_destroy": [vp],
        "m3hip_xch_set_timeout": [vp, C.c_double],
        "m3hip_device_count": [C.POINTER(i32)],
        "m3hip_par_create": [vp, C.POINTER(i32)],
        "m3hip_par_destroy": [vp, i32],
        "m3hip_par_upload": [vp, i32, i32, vp, i32],
        "m3hip_par_launch": [vp, i32, vp, vp, i32],
        "m3hip_par_result": [vp, i32, C.POINTER(C.c_double), vp, i32],
        "m3hip_get_info": [vp, C.POINTER(Info)],
        "m3hip_timing_enable": [vp, i32],
        "m3hip_timing_read": [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)],
        "m3hip_timing_reset": [vp],
        "m3hip_debug_set_bins": [vp, vp, vp, i32],
        "m3hip_debug_host_times": [vp, vp],
        "m3hip_debug_set_knobs": [vp, i32, i32, i32],
        "m3hip_debug_set_hist_replicas": [vp, i32],
        "m3hip_debug_device_span": [vp, i32, C.POINTER(C.c_double)],
        "m3hip_debug_short_brick_plan": [u64, u64, i32, i32, vp, vp],
        "m3hip_debug_layout": [i32, i32, vp, u64, vp, u64, vp, vp, u64, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    }
    for name, args in sig.items():
        if os.environ.get("M3HIP_LIB") and not hasattr(lib, name):
            continue  # A/B runs against an older build: entry points it lacks stay unbound
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    lib.m3hip_last_error.argtypes = [vp]
    lib.m3hip_last_error.restype = C.c_char_p
    lib.m3hip_version.argtypes = []
    lib.m3hip_version.restype = C.c_char_p
    _hip = lib
    return lib


def load_m3synth():
    global _synth
    if _synth is not None:
        return _synth
    path = lib_path("libm3synth.so")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: build it with `make`")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    cp, vp, u64 = C.POINTER(SynthConfig), C.c_void_p, C.c_uint64
    lib.m3synth_knots.argtypes = [cp, vp, vp]
    lib.m3synth_count.argtypes = [cp, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    lib.m3synth_monolith.argtypes = [cp, u64, u64, vp, vp, vp, vp, vp, vp, vp]
    lib.m3synth_events.argtypes = [cp, u64, u64, vp, vp, vp, vp, vp]
    lib.m3synth_osc_table.argtypes = [cp, vp]
    lib.m3synth_bin_edges.argtypes = [cp, vp]
    lib.m3synth_params.argtypes = [cp, u64, vp]
    for n in ("m3synth_knots", "m3synth_count", "m3synth_monolith", "m3synth_events", "m3synth_osc_table",
              "m3synth_bin_edges", "m3synth_params"):
        getattr(lib, n).restype = None
    _synth = lib
    return lib
