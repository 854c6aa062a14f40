"""ctypes binding of libftr_hip.so (the C ABI declared in include/ftr.h).

This is the Python side of the drop-in boundary: the reference loads its op library with
``tf.load_op_library`` (tf_fast_rnnt/python/tf_fast_rnnt/__init__.py:38-40); here the same native
entry points are plain C symbols taking device pointers, so any framework that can hand out a
device pointer and a stream (torch here; TensorFlow-ROCm through the shim in INTEGRATION.md) can
call them.  There is deliberately no CPU fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FTR_LIB_PATH") or os.path.join(_HERE, "libftr_hip.so")   # override: diagnostic builds only
_lib = None

_c_fp = ctypes.c_void_p   # const float* (device)
_c_ip = ctypes.c_void_p   # const int32_t* (device)
_c_st = ctypes.c_void_p   # hipStream_t
_i = ctypes.c_int
_f = ctypes.c_float

_SIGNATURES = {
    "ftr_abi_version": (ctypes.c_int, []),
    "ftr_package_version": (ctypes.c_char_p, []),
    "ftr_last_error": (ctypes.c_char_p, []),
    "ftr_mutual_information_workspace_floats": (ctypes.c_size_t, [_i, _i, _i]),
    "ftr_mutual_information_handoff_floats": (ctypes.c_size_t, [_i, _i, _i]),
    "ftr_mutual_information_fwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_bwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_fwd_ws_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, ctypes.c_size_t, _i, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_bwd_ws_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, ctypes.c_size_t, _i, _c_fp, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_bwd_loss_ws_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, ctypes.c_size_t, _i, _c_fp, _c_fp, _c_fp, _i, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_workspace_init": (_i, [_c_fp, ctypes.c_size_t, _i, _i, _i, _c_st]),
    "ftr_mutual_information_status": (_i, [_c_fp, ctypes.c_size_t, _i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_longlong), _c_st]),
    "ftr_cummin_i32": (_i, [_c_ip, _c_ip, _i, _i, _c_st]),
    "ftr_prune_ranges_i32": (_i, [_c_fp, _c_fp, _c_ip, _c_ip, _c_ip, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), _c_st]),
    "ftr_do_pruning_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_do_pruning_bwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_do_pruning_bwd_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i, _i, _i]),
    "ftr_do_pruning_bwd_ws_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _i, _i, _i, _i, _i, ctypes.c_void_p, ctypes.c_size_t, _c_st]),
    "ftr_pruned_logprobs_fwd_f32": (_i, [_c_fp, _c_ip, _c_ip, _c_ip, _i, ctypes.c_double, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _i, _c_st]),
    "ftr_pruned_logprobs_bwd_f32": (_i, [_c_fp, _c_ip, _c_ip, _c_ip, _i, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _i, _c_st]),
    "ftr_negated_reduce_f32": (_i, [_c_fp, _i, _i, _c_fp, _c_st]),
    "ftr_pruned_logprobs_bwd_scaled_f32": (_i, [_c_fp, _c_ip, _c_ip, _c_ip, _i, _c_fp, _c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _i, _i, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_bwd_w_scaled_f32": (_i, [_c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_bwd_am_scaled_f32": (_i, [_c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _c_fp, _c_ip, _c_ip, _i, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_rowmax_exp_f32": (_i, [_c_fp, _c_fp, _c_fp, ctypes.c_longlong, _i, _c_st]),
    "ftr_rowmax_exp_pair_f32": (_i, [_c_fp, _c_fp, _c_fp, ctypes.c_longlong, _c_fp, _c_fp, _c_fp, ctypes.c_longlong, _i, _c_st]),
    "ftr_rowmax_exp_sum_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_longlong, _i, _c_st]),
    "ftr_smoothed_logprobs_fwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_ip, _i, _f, _f, _f, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_bwd_w_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_ip, _f, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_bwd_am_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_ip, _c_ip, _i, _f, _c_fp, _c_fp, _f, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_bwd_lm_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _i, _f, _c_fp, _c_fp, _c_fp, _c_fp, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_fwd_pen_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_ip, _i, ctypes.c_double, _f, _f, _f, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_bwd_w_scaled_f32": (_i, [_c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _c_ip, _f, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_bwd_am_scaled_f32": (_i, [_c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _c_fp, _c_ip, _c_ip, _i, _f, _c_fp, _c_fp, _f, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_rowmax_exp_dot_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_longlong, _i, _c_st]),
    "ftr_rowdot_f32": (_i, [_c_fp, _c_fp, _c_fp, ctypes.c_longlong, _i, _c_st]),
    "ftr_colsum_weighted_workspace_floats": (ctypes.c_size_t, [ctypes.c_longlong, _i]),
    "ftr_colsum_weighted_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_size_t, ctypes.c_longlong, _i, _c_st]),
    "ftr_simple_logprobs_fused_supported": (_i, [_i]),
    "ftr_simple_logprobs_fused_bwd_supported": (_i, [_i, _i]),
    "ftr_normalizer_gemm_f32": (_i, [_i, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_normalizer_gemm_set_choice": (_i, [_i, _i, _i, _i, _i, _i]),
    "ftr_normalizer_gemm_choice": (_i, [_i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]),
    "ftr_simple_logprobs_fused_fwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _c_fp, _c_ip, _i, ctypes.c_double, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_fused_fwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_ip, _i, ctypes.c_double, _f, _f, _f, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_fused_bwd_am_f32": (_i, [_c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _c_fp, _c_fp, _c_ip, _c_ip, _i, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_smoothed_logprobs_fused_bwd_am_f32": (_i, [_c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _c_fp, _c_fp, _c_ip, _c_ip, _i, _f, _f, _c_fp, _c_fp, _f, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_band_supported": (_i, [_i, _i, _i]),
    "ftr_band_ranges_check_i32": (_i, [_c_ip, _c_ip, _c_ip, _i, _i, _i, _c_st]),
    "ftr_pruned_band_fwd_f32": (_i, [_c_fp, _c_ip, _c_ip, _c_ip, _i, ctypes.c_double, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_band_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_ip, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_mutual_information_band_workspace_floats": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "ftr_mutual_information_band_ws_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_ip, _c_fp, ctypes.c_size_t, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_pruned_band_bwd_scaled_f32": (_i, [_c_fp, _c_ip, _c_ip, _c_ip, _i, _c_fp, _c_fp, _c_fp, _c_fp, _i, _f, _c_fp, _i, _i, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_fwd_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _c_ip, _i, ctypes.c_double, _c_fp, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_bwd_w_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _c_fp, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_bwd_am_f32": (_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_ip, _c_ip, _i, _c_fp, _i, _i, _i, _i, _i, _c_st]),
    "ftr_simple_logprobs_bwd_lm_f32": (_i, [_c_fp, _c_fp, _c_ip, _c_fp, _c_fp, _i, _c_fp, _i, _i, _i, _c_st]),
    "ftr_selftest": (_i, [ctypes.c_void_p, _c_st]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
FTR_MI_WS_CLEAN = 1   # include/ftr.h


def lib() -> ctypes.CDLL:
    """Loads libftr_hip.so once; raises ImportError (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C tf-fast-rnnt_amd/csrc).  tf_fast_rnnt has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


class FtrError(RuntimeError):
    """A native entry point returned an FTR_ERR_* code (the reference raises tf.errors.Internal,
    tf_fast_rnnt_op.cc:114-116)."""


def check(rc: int, what: str) -> None:
    if rc != 1:
        msg = lib().ftr_last_error().decode("utf-8", "replace")
        raise FtrError(f"{what} failed with code {rc}: {msg}")


# Optional per-call hook used by bench.py: when set, every native call is bracketed by two HIP events
# recorded on the stream the kernels are launched on, keyed by the C symbol name.
_profile_hook = None


def set_profile_hook(hook) -> None:
    """hook(name) must return a context manager (or None to disable)."""
    global _profile_hook
    _profile_hook = hook


def call(name: str, *args) -> None:
    """Invoke a status-returning entry point of the C ABI and raise FtrError unless it returns 1."""
    fn = getattr(lib(), name)
    if _profile_hook is None:
        rc = fn(*args)
    else:
        with _profile_hook(name):
            rc = fn(*args)
    check(rc, name)
