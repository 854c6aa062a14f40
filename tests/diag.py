"""Test-side binding of the TEST-ONLY library csrc/_build/libftr_hip_diag.so (include/ftr_diag.h; built by `make -C
tf-fast-rnnt_amd/csrc tests`): the "plain" kernel family -- one thread per lattice row, the reference's own float32
arithmetic on the device -- which the tests use as a second, independent device implementation to compare the product
kernels with.  Nothing here is reachable from the product package."""
import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tf-fast-rnnt_amd", "csrc", "_build")
DIAG_PATH = os.path.join(BUILD, "libftr_hip_diag.so")
_i = ctypes.c_int
DIAG_SIGNATURES = {
    "ftr_set_mi_impl": (_i, [_i]),
    "ftr_get_mi_impl": (_i, []),
    "ftr_debug_stamps": (_i, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "ftr_debug_trace": (_i, [ctypes.POINTER(ctypes.c_ulonglong), _i]),
}
_cache = {}


def load(path=DIAG_PATH):
    """A ctypes handle on a diag-flavoured build (the diag library itself, the poison build, a study variant) with every
    signature of the product ABI plus the diag ones."""
    from tf_fast_rnnt import _lib
    if path not in _cache:
        assert os.path.exists(path), f"{path} is missing: make -C tf-fast-rnnt_amd/csrc tests (or __graft_entry__.build())"
        h = ctypes.CDLL(path)
        for name, (restype, argtypes) in list(_lib._SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = restype, argtypes
        _cache[path] = h
    return _cache[path]


def plain_forward_backward(px, py, boundary, need_grads=True, ans_grad=None):
    """mutual_information forward (+ backward) through the PLAIN family on the diag library: (ans, px_grad, py_grad,
    ans_grad check) as torch tensors; px / py / boundary are torch tensors on the device."""
    L = load()
    B, S, T1 = px.shape
    T = py.shape[2]
    modified = int(T1 == T)
    dev = px.device
    st = torch.cuda.current_stream(dev).cuda_stream
    px = px.contiguous(); py = py.contiguous()
    bd = None if boundary is None else boundary.to(torch.int32).contiguous()
    p = torch.empty(max(B * (S + 1) * (T + 1), 1), dtype=torch.float32, device=dev)
    ans = torch.empty((B,), dtype=torch.float32, device=dev)
    prev = L.ftr_set_mi_impl(1)
    try:
        ptr = lambda t: None if t is None else t.data_ptr()
        rc = L.ftr_mutual_information_fwd_f32(ptr(px), ptr(py), ptr(bd), ptr(p), ptr(ans), B, S, T, modified, st)
        assert rc == 1, L.ftr_last_error()
        if not need_grads:
            return ans, None, None, None
        gx = torch.empty_like(px); gy = torch.empty_like(py); pg = torch.empty_like(p)
        ag = torch.ones((B,), dtype=torch.float32, device=dev) if ans_grad is None else ans_grad.to(torch.float32).contiguous().clone()
        rc = L.ftr_mutual_information_bwd_f32(ptr(px), ptr(py), ptr(bd), ptr(p), ptr(pg), ptr(gx), ptr(gy), ptr(ag), 1, B, S, T, modified, st)
        assert rc == 1, L.ftr_last_error()
    finally:
        L.ftr_set_mi_impl(prev)
    return ans, gx, gy, ag
