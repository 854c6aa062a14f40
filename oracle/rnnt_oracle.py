"""oracle/rnnt_oracle.py -- TEST INFRASTRUCTURE ONLY.

numpy (float32) restatement of the Python layer of Samsung/tf-fast-rnnt
(``tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py`` and ``__init__.py``), on top of
the C restatement of the native kernels in ``oracle/mi_oracle.c`` (bound here with
ctypes).  It is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package never imports it.

Every function cites the reference lines it follows (paths relative to
``/root/reference/tf_fast_rnnt/python/tf_fast_rnnt/``).  TensorFlow is not
available in the build container and the reference cannot be imported
(``ModuleNotFoundError: tensorflow``), so TF ops are restated from
their documented semantics:

* ``tf.matmul``/``reduce_sum``/``cumsum``/``reduce_logsumexp`` orders are
  unspecified on GPU; here they are numpy float32 reductions (pairwise) except
  where ``mi_oracle.c`` fixes a canonical sequential order (prune-range cumsum).
* ``tf.math.nextafter(0., 1.)`` is the smallest positive float32 subnormal
  (1.4e-45).

Pinning: ``_monotonic_lower_bound`` and ``_roll_by_shifts`` are pinned against the
reference's docstring vectors (rnnt_loss.py:561-574, 823-834).  All float outputs
are PARITY UNPINNED by reference data (the reference's tests only print); they
are cross-checked by brute force / float64 autograd in tests/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

F32 = np.float32
NEXTAFTER0 = np.nextafter(np.float32(0.0), np.float32(1.0))  # tf.math.nextafter(0., 1.)


def build(force: bool = False) -> str:
    """Compile oracle/mi_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "mi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_logadd_f32.restype = ctypes.c_float
        _lib.oracle_logadd_f32.argtypes = [ctypes.c_float, ctypes.c_float]
        _lib.oracle_safe_exp_f32.restype = ctypes.c_float
        _lib.oracle_safe_exp_f32.argtypes = [ctypes.c_float]
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------------- native layer
def mi_forward(px, py, boundary=None, dtype=np.float32):
    """mutual_information_kernel (mutual_information_cuda.cu:174-422).  Returns (ans, p)."""
    px = _c(px, dtype); py = _c(py, dtype)
    B, S, T1 = px.shape
    T = py.shape[2]
    assert py.shape == (B, S + 1, T) and T1 in (T, T + 1)
    modified = int(T1 == T)
    bd = None if boundary is None else _c(boundary, np.int32)
    p = np.full((B, S + 1, T + 1), np.nan, dtype=dtype)
    ans = np.zeros((B,), dtype=dtype)
    fn = lib().oracle_mi_fwd_f32 if dtype == np.float32 else lib().oracle_mi_fwd_f64
    rc = fn(_p(px), _p(py), _p(bd), _p(p), _p(ans), B, S, T, modified)
    assert rc == 1
    return ans, p


def mi_backward(px, py, boundary, p, ans_grad=None, dtype=np.float32, overwrite_ans_grad=True):
    """mutual_information_backward_kernel (.cu:490-760).  Returns (px_grad, py_grad, ans_grad_check)."""
    px = _c(px, dtype); py = _c(py, dtype); p = _c(p, dtype)
    B, S, T1 = px.shape
    T = py.shape[2]
    modified = int(T1 == T)
    bd = None if boundary is None else _c(boundary, np.int32)
    ag = np.ones((B,), dtype=dtype) if ans_grad is None else _c(ans_grad, dtype).copy()
    p_grad = np.zeros((B, S + 1, T + 1), dtype=dtype)
    px_grad = np.zeros_like(px)   # the op memsets these (tf_fast_rnnt_op.cc:93-96)
    py_grad = np.zeros_like(py)
    fn = lib().oracle_mi_bwd_f32 if dtype == np.float32 else lib().oracle_mi_bwd_f64
    rc = fn(_p(px), _p(py), _p(bd), _p(p), _p(p_grad), _p(px_grad), _p(py_grad), _p(ag),
            int(overwrite_ans_grad), B, S, T, modified)
    assert rc == 1
    return px_grad, py_grad, ag


def mutual_information_recursion(px, py, boundary=None, calc_gradients=False, dtype=np.float32):
    """FastRNNTOpBase::Compute (tf_fast_rnnt_op.cc:48-117) + __init__.py:42-149."""
    ans, p = mi_forward(px, py, boundary, dtype)
    if not calc_gradients:
        return ans
    px_grad, py_grad, _ = mi_backward(px, py, boundary, p, None, dtype)
    return ans, (px_grad, py_grad)


def cummin(x):
    """CumminCuda (.cu:895-1012) via __init__.py:151-152."""
    x = _c(x, np.int32)
    assert x.ndim == 2
    out = np.empty_like(x)
    assert lib().oracle_cummin_i32(_p(x), _p(out), x.shape[0], x.shape[1]) == 1
    return out


def monotonic_lower_bound(x):
    """_monotonic_lower_bound (rnnt_loss.py:553-585): reverse -> cummin -> reverse."""
    x = _c(x, np.int32)
    squeeze = x.ndim == 1
    x2 = x.reshape(1, -1) if squeeze else x
    out = cummin(x2[:, ::-1].copy())[:, ::-1].copy()
    return out[0] if squeeze else out


def adjust_pruning_lower_bound(s_begin, s_range):
    """_adjust_pruning_lower_bound (rnnt_loss.py:587-641)."""
    s_begin = _c(s_begin, np.int32)
    T = s_begin.shape[1]
    s_begin = monotonic_lower_bound(s_begin)
    s_begin = -(s_begin - (s_range - 1) * np.arange(T, dtype=np.int32))
    s_begin = monotonic_lower_bound(s_begin)
    s_begin = np.clip(s_begin, 0, np.iinfo(np.int32).max)
    s_begin = -(s_begin - (s_range - 1) * np.arange(T, dtype=np.int32))
    return s_begin.astype(np.int32)


def get_rnnt_prune_ranges(px_grad, py_grad, boundary, s_range, return_raw=False):
    """get_rnnt_prune_ranges (rnnt_loss.py:647-761), canonical sequential cumsum (mi_oracle.c)."""
    px_grad = _c(px_grad, F32); py_grad = _c(py_grad, F32); boundary = _c(boundary, np.int32)
    B, S, T1 = px_grad.shape
    T = py_grad.shape[2]
    r = S + 1 if s_range > S else s_range
    ranges = np.empty((B, T, r), dtype=np.int32)
    raw = np.empty((B, T), dtype=np.int32)
    got = lib().oracle_prune_ranges(_p(px_grad), _p(py_grad), _p(boundary), _p(ranges), _p(raw),
                                    B, S, T, T1, int(s_range))
    assert got == r
    return (ranges, raw) if return_raw else ranges


def get_rnnt_prune_ranges_numpy(px_grad, py_grad, boundary, s_range):
    """Same function written op-by-op like the reference (rnnt_loss.py:695-761); numpy cumsum along S
    is sequential for a strided axis, so this agrees bit-for-bit with the C version (tested)."""
    px_grad = _c(px_grad, F32); py_grad = _c(py_grad, F32); boundary = _c(boundary, np.int32)
    B, S, T1 = px_grad.shape
    T = py_grad.shape[-1]
    S1 = S + 1
    if s_range > S:
        s_range = S + 1
    cumsum = np.zeros((B, S1 + 1, T), dtype=F32)
    acc = np.zeros((B, T), dtype=F32)
    for s in range(S1):                     # sequential f32, the canonical order
        acc = (acc + py_grad[:, s, :]).astype(F32)
        cumsum[:, s + 1, :] = acc
    blk_sum_grad = cumsum[:, s_range:, :] - cumsum[:, : S1 - s_range + 1, :]
    px_grad_pad = np.concatenate((np.zeros((B, 1, T1), dtype=F32), px_grad), axis=1)
    final_grad = blk_sum_grad - px_grad_pad[:, : S1 - s_range + 1, :T]
    s_begin = np.argmax(final_grad, axis=1).astype(np.int32)          # first maximum
    mask = np.arange(T, dtype=np.int32)[None, :] < (boundary[:, 3].reshape(B, 1) - 1)
    s_begin_padding = np.clip(boundary[:, 2].reshape(B, 1) - s_range + 1, 0, None)
    s_begin = np.where(mask, s_begin, s_begin_padding).astype(np.int32)
    s_begin = adjust_pruning_lower_bound(s_begin, 2 if T1 == T else s_range)
    return (s_begin[:, :, None] + np.arange(s_range, dtype=np.int32)).astype(np.int32)


def do_rnnt_pruning(am, lm, ranges):
    """do_rnnt_pruning (rnnt_loss.py:763-812)."""
    am = _c(am, F32); lm = _c(lm, F32); ranges = _c(ranges, np.int32)
    B, T, r = ranges.shape
    C = lm.shape[2]
    am_p = np.empty((B, T, r, C), dtype=F32)
    lm_p = np.empty((B, T, r, C), dtype=F32)
    assert lib().oracle_do_pruning(_p(am), _p(lm), _p(ranges), _p(am_p), _p(lm_p), B, T,
                                   lm.shape[1], C, r) == 1
    return am_p, lm_p


def roll_by_shifts(src, shifts):
    """_roll_by_shifts (rnnt_loss.py:814-851): out[b,t,i] = src[b,t,(i - shifts[b,t]) % S]."""
    src = np.asarray(src)
    B, T, S = src.shape
    index = (np.arange(S)[None, None, :] - np.asarray(shifts).reshape(B, T, 1)) % S
    return np.take_along_axis(src, index, axis=2)


# --------------------------------------------------------------------------- px/py builders
def fix_for_boundary(px, boundary):
    """fix_for_boundary (rnnt_loss.py:28-61): px[b, :, boundary[b,3]] = -inf."""
    if boundary is None:
        return px
    px = px.copy()
    for b in range(px.shape[0]):
        px[b, :, int(boundary[b, 3])] = -np.inf
    return px


def _normalizers(lm, am):
    """rnnt_loss.py:175-186."""
    am_max = am.max(axis=2, keepdims=True)
    lm_max = lm.max(axis=2, keepdims=True)
    am_probs = np.exp(am - am_max, dtype=F32)
    lm_probs = np.exp(lm - lm_max, dtype=F32)
    prod = np.matmul(lm_probs, am_probs.transpose(0, 2, 1)).astype(F32)
    normalizers = np.log(prod + NEXTAFTER0, dtype=F32)
    normalizers = (normalizers + lm_max + am_max.transpose(0, 2, 1)).astype(F32)
    return normalizers, am_max, lm_max, am_probs, lm_probs


def get_rnnt_logprobs(lm, am, symbols, termination_symbol, rnnt_type="regular", boundary=None):
    """get_rnnt_logprobs (rnnt_loss.py:63-223)."""
    assert rnnt_type in ("regular", "modified", "constrained")
    lm = _c(lm, F32); am = _c(am, F32); symbols = np.asarray(symbols)
    B, T, C = am.shape
    S = lm.shape[1] - 1
    normalizers, *_ = _normalizers(lm, am)                                  # [B,S+1,T]
    px_am = np.take_along_axis(am.transpose(0, 2, 1), symbols[:, :, None].astype(np.int64), axis=1)  # [B,S,T]
    if rnnt_type == "regular":
        px_am = np.concatenate((px_am, np.full((B, S, 1), -np.inf, dtype=F32)), axis=2)
    px_lm = np.take_along_axis(lm[:, :S, :], symbols[:, :, None].astype(np.int64), axis=2)  # [B,S,1]
    px = (px_am + px_lm).astype(F32)
    if rnnt_type == "regular":
        # rnnt_loss.py:211 pads normalizers with a zero column then slices [:S]
        px = px - np.concatenate((normalizers, np.zeros((B, S + 1, 1), dtype=F32)), axis=2)[:, :S, :]
    else:
        # the reference line :211 raises a shape error for non-regular types (reference
        # bugs); the intended arithmetic (upstream k2) subtracts the unpadded normalizers.
        px = px - normalizers[:, :S, :]
    py_am = am[:, :, termination_symbol][:, None, :]
    py_lm = lm[:, :, termination_symbol][:, :, None]
    py = (py_am + py_lm - normalizers).astype(F32)
    if rnnt_type == "regular":
        px = fix_for_boundary(px, boundary)
    elif rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return px.astype(F32), py


def get_rnnt_logprobs_smoothed(lm, am, symbols, termination_symbol, lm_only_scale=0.1,
                               am_only_scale=0.1, boundary=None, rnnt_type="regular"):
    """get_rnnt_logprobs_smoothed (rnnt_loss.py:1132-1367)."""
    lm = _c(lm, F32); am = _c(am, F32); symbols = np.asarray(symbols)
    B, T, C = am.shape
    S = lm.shape[1] - 1
    normalizers, am_max, lm_max, am_probs, lm_probs = _normalizers(lm, am)
    lmonly_normalizers = lm_probs.sum(axis=2, keepdims=True, dtype=F32)               # :1276-1278
    unigram_lm = (np.mean(lm_probs / lmonly_normalizers, axis=(0, 1), keepdims=True, dtype=F32)
                  + NEXTAFTER0).astype(F32)                                           # :1279-1280  [1,1,C]
    amonly_normalizers = (np.log(np.matmul(am_probs.reshape(-1, C), unigram_lm.reshape(C)),
                                 dtype=F32).reshape(B, T, 1) + am_max)                # :1281-1285
    amonly_normalizers = amonly_normalizers.transpose(0, 2, 1).astype(F32)            # [B,1,T]
    unigram_lm = np.log(unigram_lm, dtype=F32)
    lmonly_normalizers = (np.log(lmonly_normalizers, dtype=F32) + lm_max).astype(F32)  # [B,S+1,1]

    sym = symbols.astype(np.int64)
    px_am = np.take_along_axis(am.transpose(0, 2, 1), sym[:, :, None], axis=1)        # [B,S,T]
    regular = rnnt_type == "regular"
    if regular:
        px_am = np.concatenate((px_am, np.full((B, S, 1), -np.inf, dtype=F32)), axis=2)
    px_lm = np.take_along_axis(lm[:, :S, :], sym[:, :, None], axis=2)                 # [B,S,1]
    px_lm_unigram = unigram_lm.reshape(-1)[sym][:, :, None]                           # [B,S,1]
    px = (px_am + px_lm).astype(F32)
    if regular:
        px = px - np.concatenate((normalizers, np.zeros((B, S + 1, 1), dtype=F32)), axis=2)[:, :S, :]
        px_amonly = (px_am + px_lm_unigram) - np.concatenate(
            (amonly_normalizers, np.zeros((B, 1, 1), dtype=F32)), axis=2)             # :1326-1330
    else:
        px = px - normalizers[:, :S, :]
        px_amonly = (px_am + px_lm_unigram) - amonly_normalizers
    px_lmonly = px_lm - lmonly_normalizers[:, :S, :]                                  # :1331

    py_am = am[:, :, termination_symbol][:, None, :]
    py_lm = lm[:, :, termination_symbol][:, :, None]
    py = py_am + py_lm - normalizers
    py_lm_unigram = unigram_lm[0, 0, termination_symbol]
    py_amonly = py_am + py_lm_unigram - amonly_normalizers                            # [B,1,T]
    py_lmonly = py_lm - lmonly_normalizers                                            # [B,S+1,1]

    combined_scale = 1.0 - lm_only_scale - am_only_scale
    if lm_only_scale == 0.0:
        lm_only_scale = 1.0e-20
    if am_only_scale == 0.0:
        am_only_scale = 1.0e-20
    cs, ls, as_ = F32(combined_scale), F32(lm_only_scale), F32(am_only_scale)
    px_interp = (px * cs + px_lmonly * ls + px_amonly * as_).astype(F32)
    py_interp = (py * cs + py_lmonly * ls + py_amonly * as_).astype(F32)
    if regular:
        px_interp = fix_for_boundary(px_interp, boundary)
    elif rnnt_type == "constrained":
        px_interp = px_interp + py_interp[:, 1:, :]
    return px_interp.astype(F32), py_interp.astype(F32)


def _logsumexp(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return (np.squeeze(m, axis) + np.log(np.exp(x - m, dtype=F32).sum(axis=axis, dtype=F32), dtype=F32)).astype(F32)


def get_rnnt_logprobs_joint(logits, symbols, termination_symbol, boundary=None, rnnt_type="regular"):
    """get_rnnt_logprobs_joint (rnnt_loss.py:340-452).  logits [B,T,S+1,C]."""
    logits = _c(logits, F32); symbols = np.asarray(symbols)
    B, T, S1, C = logits.shape
    S = S1 - 1
    normalizers = _logsumexp(logits, 3).transpose(0, 2, 1)                     # [B,S+1,T]
    idx = np.broadcast_to(symbols.astype(np.int64).reshape(B, 1, S, 1), (B, T, S, 1))
    px = np.take_along_axis(logits[:, :, :S, :], idx, axis=3)[..., 0].transpose(0, 2, 1)  # [B,S,T]
    if rnnt_type == "regular":
        px = np.concatenate((px, np.full((B, S, 1), -np.inf, dtype=F32)), axis=2)
        px = px - np.concatenate((normalizers, np.zeros((B, S + 1, 1), dtype=F32)), axis=2)[:, :S, :]
    else:
        px = px - normalizers[:, :S, :]
    py = logits[:, :, :, termination_symbol].transpose(0, 2, 1) - normalizers
    if rnnt_type == "regular":
        px = fix_for_boundary(px, boundary)
    elif rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return px.astype(F32), py.astype(F32)


def get_rnnt_logprobs_pruned(logits, symbols, ranges, termination_symbol, boundary, rnnt_type="regular"):
    """get_rnnt_logprobs_pruned (rnnt_loss.py:853-1020), op by op."""
    logits = _c(logits, F32); symbols = np.asarray(symbols); ranges = np.asarray(ranges)
    B, T, s_range, C = logits.shape
    S = symbols.shape[1]
    normalizers = _logsumexp(logits, 3)                                        # :942   [B,T,r]
    symbols_with_terminal = np.concatenate(
        (symbols, np.full((B, 1), termination_symbol, dtype=symbols.dtype)), axis=1)   # :943-951
    pruned_symbols = np.take_along_axis(
        np.broadcast_to(symbols_with_terminal[:, None, :], (B, T, S + 1)), ranges.astype(np.int64), axis=2)
    px = np.take_along_axis(logits, pruned_symbols[..., None].astype(np.int64), axis=3)[..., 0]
    px = px - normalizers                                                      # :965
    px = np.concatenate((px, np.full((B, T, S + 1 - s_range), -np.inf, dtype=F32)), axis=2)
    px = roll_by_shifts(px, ranges[:, :, 0])[:, :, :S]                         # :980
    px = px.transpose(0, 2, 1)
    if rnnt_type == "regular":
        px = np.concatenate((px, np.full((B, S, 1), -np.inf, dtype=F32)), axis=2)
    py = logits[:, :, :, termination_symbol] - normalizers                     # :995-996
    py = np.concatenate((py, np.full((B, T, S + 1 - s_range), -np.inf, dtype=F32)), axis=2)
    py = roll_by_shifts(py, ranges[:, :, 0]).transpose(0, 2, 1)                # :1011-1013
    if rnnt_type == "regular":
        px = fix_for_boundary(px, boundary)
    elif rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return np.ascontiguousarray(px, dtype=F32), np.ascontiguousarray(py, dtype=F32)


# --------------------------------------------------------------------------- loss drivers
def _delay_penalty(px, boundary, rnnt_type, delay_penalty):
    """rnnt_loss.py:305-321 (float64 arithmetic, cast to f32)."""
    if not delay_penalty > 0.0:
        return px
    B, S, T0 = px.shape
    T = T0 if rnnt_type != "regular" else T0 - 1
    if boundary is None:
        offset = np.full((B,), (T - 1) / 2, dtype=np.float64)
    else:
        offset = (np.asarray(boundary)[:, 3].astype(np.float64) - 1) / 2
    penalty = offset.reshape(B, 1, 1) - np.arange(T0, dtype=np.float64).reshape(1, 1, T0)
    penalty = penalty * delay_penalty
    return (px + penalty.astype(F32)).astype(F32)


def _reduce(negated_loss, reduction):
    if reduction == "none":
        return -negated_loss
    if reduction == "mean":
        # rnnt_loss_simple's "mean" branch is a NameError in the reference (rnnt_loss.py:331);
        # the other drivers use reduce_mean (:544,1124,1487), restated here for all.
        return -np.mean(negated_loss, dtype=F32)
    if reduction == "sum":
        return -np.sum(negated_loss, dtype=F32)
    raise ValueError(f"reduction should be ('none' | 'mean' | 'sum'), given {reduction}")


def _drive(px, py, boundary, reduction, calc_gradients, dtype=np.float32):
    out = mutual_information_recursion(px, py, boundary, calc_gradients, dtype)
    negated = out[0] if calc_gradients else out
    loss = _reduce(negated, reduction)
    return (loss, out[1]) if calc_gradients else loss


def rnnt_loss_simple(lm, am, symbols, termination_symbol, boundary=None, rnnt_type="regular",
                     delay_penalty=0.0, reduction="mean", calc_gradients=False):
    """rnnt_loss_simple (rnnt_loss.py:225-338)."""
    px, py = get_rnnt_logprobs(lm, am, symbols, termination_symbol, rnnt_type, boundary)
    px = _delay_penalty(px, boundary, rnnt_type, delay_penalty)
    return _drive(px, py, boundary, reduction, calc_gradients)


def rnnt_loss_smoothed(lm, am, symbols, termination_symbol, lm_only_scale=0.1, am_only_scale=0.1,
                       boundary=None, rnnt_type="regular", delay_penalty=0.0, reduction="mean",
                       calc_gradients=False):
    """rnnt_loss_smoothed (rnnt_loss.py:1369-1494)."""
    px, py = get_rnnt_logprobs_smoothed(lm, am, symbols, termination_symbol, lm_only_scale,
                                        am_only_scale, boundary, rnnt_type)
    px = _delay_penalty(px, boundary, rnnt_type, delay_penalty)
    return _drive(px, py, boundary, reduction, calc_gradients)


def rnnt_loss(logits, symbols, termination_symbol, boundary=None, rnnt_type="regular",
              delay_penalty=0.0, reduction="mean", calc_gradients=False):
    """rnnt_loss (rnnt_loss.py:454-551)."""
    px, py = get_rnnt_logprobs_joint(logits, symbols, termination_symbol, boundary, rnnt_type)
    px = _delay_penalty(px, boundary, rnnt_type, delay_penalty)
    return _drive(px, py, boundary, reduction, calc_gradients)


def rnnt_loss_pruned(logits, symbols, ranges, termination_symbol, boundary=None, rnnt_type="regular",
                     delay_penalty=0.0, reduction="mean", calc_gradients=False):
    """rnnt_loss_pruned (rnnt_loss.py:1022-1130); returns the loss only."""
    px, py = get_rnnt_logprobs_pruned(logits, symbols, ranges, termination_symbol, boundary, rnnt_type)
    px = _delay_penalty(px, boundary, rnnt_type, delay_penalty)
    out = mutual_information_recursion(px, py, boundary, calc_gradients)
    negated = out[0] if calc_gradients else out
    return _reduce(negated, reduction)


def rnnt_loss_pruned_grad(logits, symbols, ranges, termination_symbol, boundary, rnnt_type="regular",
                          delay_penalty=0.0, reduction="mean", dtype=np.float32):
    """d rnnt_loss_pruned / d logits as TF autodiff produces it: the custom-op gradient
    (__init__.py:154-162: ans_grad * px_grad / py_grad) chained through get_rnnt_logprobs_pruned
    (gather of the band, minus logsumexp).  Returns (loss, grad [B,T,r,C])."""
    logits = _c(logits, F32); symbols = _c(symbols, np.int32); ranges = _c(ranges, np.int32)
    B, T, r, C = logits.shape
    S = symbols.shape[1]
    px, py = get_rnnt_logprobs_pruned(logits, symbols, ranges, termination_symbol, boundary, rnnt_type)
    px = _delay_penalty(px, boundary, rnnt_type, delay_penalty)
    # dtype=float64 runs the recursion (the long dependent chain) in double: the "exact" comparison point
    ans, (px_grad, py_grad) = mutual_information_recursion(px, py, boundary, True, dtype)
    loss = _reduce(ans.astype(F32), reduction)
    px_grad = px_grad.astype(F32); py_grad = py_grad.astype(F32)
    scale = {"none": -1.0, "sum": -1.0, "mean": -1.0 / B}[reduction]   # d loss / d ans[b]
    # band gradients: px[b,s,t] came from logits[b,t,s-s0,:] for s0 <= s < s0+r (s < S, t < T)
    s0 = ranges[:, :, 0]
    gx = np.zeros((B, T, r), dtype=F32)
    gy = np.zeros((B, T, r), dtype=F32)
    for k in range(r):
        s = s0 + k                                                      # [B,T]
        bb, tt = np.meshgrid(np.arange(B), np.arange(T), indexing="ij")
        ok_x = s < S
        gx[:, :, k] = np.where(ok_x, px_grad[bb, np.minimum(s, S - 1), tt], 0.0)
        gy[:, :, k] = py_grad[bb, np.minimum(s, S), tt]
    if rnnt_type == "constrained":
        raise NotImplementedError
    gx *= F32(scale); gy *= F32(scale)
    lse = _logsumexp(logits, 3)
    g = np.empty_like(logits)
    assert lib().oracle_pruned_band_bwd(_p(logits), _p(symbols), _p(ranges), int(termination_symbol),
                                        _p(_c(lse, F32)), _p(_c(gx, F32)), _p(_c(gy, F32)), _p(g),
                                        B, T, S, C, r) == 1
    return loss, g


def pruned_band_fwd(logits, symbols, ranges, termination_symbol):
    logits = _c(logits, F32); symbols = _c(symbols, np.int32); ranges = _c(ranges, np.int32)
    B, T, r, C = logits.shape
    S = symbols.shape[1]
    lse = np.empty((B, T, r), dtype=F32); pxb = np.empty_like(lse); pyb = np.empty_like(lse)
    assert lib().oracle_pruned_band_fwd(_p(logits), _p(symbols), _p(ranges), int(termination_symbol),
                                        _p(lse), _p(pxb), _p(pyb), B, T, S, C, r) == 1
    return lse, pxb, pyb


# --------------------------------------------------------------------------- independent checks
def brute_force_mi(px, py, boundary=None, modified=False):
    """Sum over all monotonic paths by explicit enumeration, float64 -- independent of the recursion.
    Only for tiny lattices.  Returns ans [B]."""
    px = np.asarray(px, dtype=np.float64); py = np.asarray(py, dtype=np.float64)
    B, S, T1 = px.shape
    T = py.shape[2]
    out = np.zeros((B,), dtype=np.float64)
    for b in range(B):
        sb, tb, se, te = (0, 0, S, T) if boundary is None else [int(v) for v in boundary[b]]
        total = [0.0]

        def rec(s, t, logp):
            if s == se and t == te:
                total[0] += np.exp(logp)
                return
            if not modified:
                if s < se and t <= te and np.isfinite(px[b, s, t]):
                    rec(s + 1, t, logp + px[b, s, t])
            else:
                if s < se and t < te and np.isfinite(px[b, s, t]):
                    rec(s + 1, t + 1, logp + px[b, s, t])
            if t < te and np.isfinite(py[b, s, t]):
                rec(s, t + 1, logp + py[b, s, t])

        rec(sb, tb, 0.0)
        out[b] = np.log(total[0]) if total[0] > 0 else -np.inf
    return out


def mi_fwd_bwd_timed(px, py, boundary, threads: int):
    """cpu_baseline helper: one fwd+bwd pass of the C restatement over the batch; returns seconds."""
    import time
    px = _c(px, F32); py = _c(py, F32)
    B, S, T1 = px.shape
    T = py.shape[2]
    bd = None if boundary is None else _c(boundary, np.int32)
    p = np.empty((B, S + 1, T + 1), dtype=F32); pg = np.empty_like(p)
    pxg = np.zeros_like(px); pyg = np.zeros_like(py)
    ans = np.empty((B,), dtype=F32); ag = np.ones((B,), dtype=F32)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(threads))
    except OSError:
        pass
    t0 = time.perf_counter()
    lib().oracle_mi_fwd_bwd_f32_mt(_p(px), _p(py), _p(bd), _p(p), _p(pg), _p(pxg), _p(pyg), _p(ans),
                                   _p(ag), B, S, T, int(T1 == T))
    return time.perf_counter() - t0, ans
