"""Host wrappers of the native ops: ``mutual_information_recursion`` and ``cummin``.

Mirrors tf_fast_rnnt/python/tf_fast_rnnt/__init__.py:42-162 of the reference (op call + registered
gradient) and the op kernel it drives, ``FastRNNTOpBase::Compute``
(tf_fast_rnnt/python/csrc/tf_fast_rnnt_op.cc:48-117): allocate the ``p`` workspace and the outputs,
run the forward, and -- when gradients are wanted -- the backward seeded with ones on the same stream.
Differences from the reference, all deliberate (SURVEY.md section 7 "hard parts"):

* tensors are torch tensors on a HIP device; work is enqueued on torch's current stream and the
  host is never blocked (the reference calls cudaStreamSynchronize per op, tf_fast_rnnt_op.cc:113);
* ``boundary=None`` works (defaults to (0,0,S,T) in the kernels); the reference advertises it but
  dereferences the tensor unconditionally (mutual_information_cuda.cu:259-260);
* gradients are computed whenever autograd needs them, not only when ``calc_gradients`` is set (the
  reference back-propagates an uninitialised buffer in that case, tf_fast_rnnt_op.cc:83-98);
* for a modified-type ``px`` ([B,S,T]) ``px_grad`` has the shape of ``px`` (the reference always
  allocates [B,S,T+1], tf_fast_rnnt_op.cc:84, which cannot be multiplied into the gradient).
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import _lib


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"tf_fast_rnnt: `{name}` is on {t.device}; the native ops run on a HIP GPU only "
            "(the reference registers DEVICE_GPU kernels only, tf_fast_rnnt_op.cc:131,164) and there is no CPU fallback")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _as_boundary(boundary, B: int, device) -> Optional[torch.Tensor]:
    if boundary is None:
        return None
    boundary = torch.as_tensor(boundary, device=device)
    if boundary.dtype != torch.int32:
        boundary = boundary.to(torch.int32)
    if tuple(boundary.shape) != (B, 4):
        raise ValueError(f"boundary must have shape ({B}, 4), got {tuple(boundary.shape)}")
    return boundary.contiguous()


class _Workspace:
    """ONE cached forward->backward workspace per (device, stream), sized by CAPACITY, not by shape: a training loop pads
    every batch to its own maximum, so (B, S, T) change from step to step and a cache keyed on the exact shape would
    allocate and initialise a workspace per step.  The native layout anchors the hand-off region of a launch at the END of
    the buffer it is given and grows everything else from the front (include/ftr.h, FTR_MI_WS_CLEAN), so one buffer of
    (largest data part seen) + (largest hand-off part seen) floats serves every shape seen so far; its tail is zeroed when
    the buffer is (re)allocated -- on growth only -- and every launch leaves its hand-off region zero again, so every
    launch gets FTR_MI_WS_CLEAN and no memset node.  Safe because every user of the workspace in this package runs the
    forward and the backward launch back to back on one stream (the occupancies, not the workspace, are what autograd
    keeps).  Under stream capture the cache is bypassed (a graph owns its allocations).

    If a launch ever times out waiting for a producer (ftr_mutual_information_status bit 0: cannot happen unless a
    workgroup never ran) its `ans` is NaN and the hand-off region is left dirty: call ``clear_workspace_cache()`` (or
    ``check_workspace_status()``, which synchronises, reads the status word and drops the dirty buffers)."""
    _cache = {}
    inits = 0      # (re)allocations so far: the tests assert that a ragged loop does not allocate per step

    def __init__(self):
        self.ws = None; self.cap = 0; self.data_max = 0; self.handoff_max = 0; self.last = None

    @classmethod
    def get(cls, L, device, B, S, T):
        st = torch.cuda.current_stream(device)
        total = L.ftr_mutual_information_workspace_floats(B, S, T)
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(total, dtype=torch.float32, device=device), total, 0
        key = (device.index, st.cuda_stream)
        w = cls._cache.get(key)
        if w is None:
            if len(cls._cache) >= 4:           # a handful of (device, stream) pairs; do not hoard HBM beyond that
                cls._cache.pop(next(iter(cls._cache)))
            w = cls._cache[key] = _Workspace()
        handoff = L.ftr_mutual_information_handoff_floats(B, S, T)
        data = total - handoff
        if w.ws is None or data > w.data_max or handoff > w.handoff_max:
            # grow with some headroom so that a slowly rising maximum does not reallocate every few steps
            w.data_max = max(w.data_max, data + data // 8 if w.ws is not None else data)
            w.handoff_max = max(w.handoff_max, handoff + handoff // 8 if w.ws is not None else handoff)
            w.cap = w.data_max + w.handoff_max + 8
            w.ws = None                        # release before allocating the larger one
            w.ws = torch.empty(w.cap, dtype=torch.float32, device=device)
            w.ws[w.cap - w.handoff_max - 8:].zero_()     # every shape's hand-off region lies inside this tail
            cls.inits += 1
        w.last = (B, S, T)
        return w.ws, w.cap, _lib.FTR_MI_WS_CLEAN


def clear_workspace_cache() -> None:
    """Drops the cached recursion workspaces (they are re-created on demand)."""
    _Workspace._cache.clear()


def check_workspace_status() -> int:
    """Synchronises and reads the sticky status word of every cached workspace (ftr_mutual_information_status): returns
    the OR of them and drops the workspaces that report a problem, so that the next launch starts from a clean one.
    0 = fine.  Diagnostic: a non-zero value means a launch gave up waiting for a producer and answered NaN."""
    import ctypes
    L = _lib.lib()
    bad = 0
    for key, w in list(_Workspace._cache.items()):
        if w.ws is None or w.last is None:
            continue
        B, S, T = w.last
        st = ctypes.c_int(-1)
        with torch.cuda.device(w.ws.device):
            _lib.call("ftr_mutual_information_status", w.ws.data_ptr(), w.cap, B, S, T, ctypes.byref(st), None, key[1])
        if st.value != 0:
            bad |= st.value
            del _Workspace._cache[key]
    return bad


def mi_forward_backward(px: torch.Tensor, py: torch.Tensor, boundary: Optional[torch.Tensor],
                        need_grads: bool, ans_grad: Optional[torch.Tensor] = None,
                        return_ans_grad_check: bool = False, ans_grad_is_one: bool = False, loss_code: Optional[int] = None):
    """FastRNNTOpBase::Compute on raw tensors (no autograd).  Returns (ans, px_grad|None, py_grad|None[, check]).
    ``loss_code`` (0 none / 1 mean / 2 sum, with ``ans_grad_is_one`` and ``need_grads``): the backward launch also writes the
    negated / reduced loss (ftr_mutual_information_bwd_loss_ws_f32) and it is returned as a fourth element."""
    _require_gpu(px, "px"); _require_gpu(py, "py")
    if px.dtype != torch.float32 or py.dtype != torch.float32:
        raise TypeError("px and py must be float32 (op registration: tf_fast_rnnt_op.cc:27-34)")
    if px.dim() != 3 or py.dim() != 3:
        raise ValueError("px and py must be 3-dimensional")
    B, S, T1 = px.shape
    T = py.shape[2]
    if T1 not in (T, T + 1):
        raise ValueError(f"px.shape[-1]={T1} must be T or T+1 with T=py.shape[-1]={T}")
    if tuple(py.shape) != (B, S + 1, T):
        raise ValueError(f"py must have shape {(B, S + 1, T)}, got {tuple(py.shape)}")
    modified = int(T1 == T)
    px = px.contiguous(); py = py.contiguous()
    boundary = _as_boundary(boundary, B, px.device)
    L = _lib.lib()
    with torch.cuda.device(px.device):
        st = _stream_ptr(px)
        ws, ws_floats, flags = _Workspace.get(L, px.device, B, S, T)
        ans = torch.empty((B,), dtype=torch.float32, device=px.device)
        _lib.call("ftr_mutual_information_fwd_ws_f32", _ptr(px), _ptr(py), _ptr(boundary), _ptr(ws), ws_floats, flags,
                                                       _ptr(ans), B, S, T, modified, st)
        if not need_grads:
            return (ans, None, None, None) if return_ans_grad_check else (ans, None, None)
        px_grad = torch.empty_like(px)
        py_grad = torch.empty_like(py)
        # ans_grad := 1 like the op (tf_fast_rnnt_op.cc:104-107); the kernel overwrites it with
        # p_grad[s_begin,t_begin] as the reference's self-check does (overwrite_ans_grad = true, :109-110)
        if ans_grad_is_one:
            ag, overwrite = None, 0          # NULL ans_grad = ones, no self-check write-back: one launch less
        else:
            ag = torch.ones((B,), dtype=torch.float32, device=px.device) if ans_grad is None else ans_grad.to(torch.float32).contiguous().clone()
            overwrite = 1
        if loss_code is not None and ans_grad_is_one and not return_ans_grad_check:
            loss = torch.empty((B,) if loss_code == 0 else (), dtype=torch.float32, device=px.device)
            _lib.call("ftr_mutual_information_bwd_loss_ws_f32", _ptr(px), _ptr(py), _ptr(boundary), _ptr(ws), ws_floats, flags,
                      _ptr(px_grad), _ptr(py_grad), _ptr(ans), int(loss_code), _ptr(loss), B, S, T, modified, st)
            return ans, px_grad, py_grad, loss
        # p_grad = NULL: the reference's [B,S+1,T+1] gradient lattice (tf_fast_rnnt_op.cc:90-91) never exists here
        _lib.call("ftr_mutual_information_bwd_ws_f32", _ptr(px), _ptr(py), _ptr(boundary), _ptr(ws), ws_floats, flags,
                                                       None, _ptr(px_grad), _ptr(py_grad), _ptr(ag), overwrite,
                                                       B, S, T, modified, st)
    return (ans, px_grad, py_grad, ag) if return_ans_grad_check else (ans, px_grad, py_grad)


class _MutualInformation(torch.autograd.Function):
    """Op "FastRNNTLoss" + its registered gradient (__init__.py:154-162)."""

    @staticmethod
    def forward(ctx, px, py, boundary, calc_gradients):
        need = bool(calc_gradients) or px.requires_grad or py.requires_grad
        ans, px_grad, py_grad = mi_forward_backward(px.detach(), py.detach(), boundary, need)
        if need:
            ctx.save_for_backward(px_grad, py_grad)
        ctx.have_grads = need
        if px_grad is None:
            px_grad = torch.zeros_like(px)
            py_grad = torch.zeros_like(py)
        ctx.mark_non_differentiable(px_grad, py_grad)
        ctx.set_materialize_grads(False)          # no zero tensors for the two occupancy outputs in backward
        return ans, px_grad, py_grad

    @staticmethod
    def backward(ctx, g_ans, _g1, _g2):
        if not ctx.have_grads:
            raise RuntimeError("mutual_information_recursion: backward without saved occupancies")
        px_grad, py_grad = ctx.saved_tensors
        if g_ans is None:
            return None, None, None, None
        g = g_ans.reshape(-1, 1, 1)            # _RNNTLossGrad: ans_grad * gradpx, ans_grad * gradpy
        return g * px_grad, g * py_grad, None, None


def mutual_information_recursion(
    px: torch.Tensor,
    py: torch.Tensor,
    boundary: Optional[torch.Tensor] = None,
    calc_gradients: bool = False,
) -> Union[Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], torch.Tensor]:
    """Same contract as the reference's ``mutual_information_recursion`` (__init__.py:42-149).

    px: [B,S,T+1] (regular) or [B,S,T] (modified); py: [B,S+1,T]; boundary: int32 [B,4] rows
    (s_begin, t_begin, s_end, t_end) or None.  Returns ``ans`` [B] with
    ``p[b,s,t] = log_add(p[b,s-1,t(+off)] + px[b,s-1,t(+off)], p[b,s,t-1] + py[b,s,t-1])``,
    ``ans[b] = p[b,s_end,t_end]``; with ``calc_gradients`` also ``(px_grad, py_grad)``, the occupation
    probabilities (gradient of ``ans.sum()``).  Differentiable w.r.t. px and py.
    """
    ans, px_grad, py_grad = _MutualInformation.apply(px, py, boundary, calc_gradients)
    return (ans, (px_grad, py_grad)) if calc_gradients else ans


def cummin(x: torch.Tensor) -> torch.Tensor:
    """Op "Cummin" (__init__.py:151-152; tf_fast_rnnt_op.cc:135-165): inclusive prefix-min along the
    last axis of an int32 [rows, cols] matrix."""
    _require_gpu(x, "x")
    if x.dim() != 2:
        raise ValueError("cummin expects a 2-D tensor")
    if x.dtype != torch.int32:
        raise TypeError("cummin expects int32 (op registration: tf_fast_rnnt_op.cc:36-38)")
    x = x.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.call("ftr_cummin_i32", _ptr(x), _ptr(out), x.shape[0], x.shape[1], _stream_ptr(x))
    return out
