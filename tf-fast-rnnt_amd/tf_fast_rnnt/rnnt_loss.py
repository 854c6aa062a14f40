"""The Python op surface of tf_fast_rnnt on MI355X: same names, keyword signatures, defaults and return
conventions as the reference's ``tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py`` (cited per function
as ``rnnt_loss.py:<lines>``), on torch tensors resident in HBM.

What runs where:
* ``mutual_information_recursion`` (every loss ends there), ``get_rnnt_prune_ranges``,
  ``do_rnnt_pruning``, ``get_rnnt_logprobs_pruned`` / ``rnnt_loss_pruned``: hand-written HIP behind the
  C ABI (include/ftr.h);
* ``get_rnnt_logprobs`` / ``rnnt_loss_simple``: hand-written HIP prologue, epilogue and backward kernels around
  the normaliser GEMM (rocBLAS behind ftr_normalizer_gemm_f32);
* ``get_rnnt_logprobs_smoothed`` / ``rnnt_loss_smoothed``: the same native kernels with the LM-only / AM-only
  terms folded in; only the [C]- and [rows]-sized batch statistics (unigram mean, two matvecs) are torch ops;
* ``get_rnnt_logprobs_joint`` / ``rnnt_loss`` (unpruned, joiner logits [B,T,S+1,C]): the pruned builder's kernels
  with identity ranges (s_range = S+1).

Reference bugs that are NOT reproduced: ``rnnt_loss_simple(reduction="mean")``
raises NameError there (rnnt_loss.py:331) -- here it is the mean; ``boundary=None`` works; the
non-regular ``rnnt_type`` values do not hit the shape error of rnnt_loss.py:211/440/1324 (the
normalisers are used unpadded, as upstream k2 does).
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import os
import torch

from . import _lib
from .mutual_information import (_as_boundary, _ptr, _require_gpu, _stream_ptr, cummin,
                                 mi_forward_backward, mutual_information_recursion)

_NEG_INF = float("-inf")
# tf.math.nextafter(0., 1.) : smallest positive float32 subnormal (rnnt_loss.py:181,1272,1280)
_TINY = 1.401298464324817e-45


def tune_normalizer_gemms(enable: bool = True, filename: Optional[str] = None, search: bool = True, when: str = "second") -> None:
    """The dense contractions of the simple / smoothed builders that stay library f32 GEMMs (the two transposes of the
    normaliser product in the backward; also the forward one when C % 4 != 0, rnnt_loss.py:180-182) go through
    ``ftr_normalizer_gemm_f32`` (csrc/normalizer_gemm.hip): rocBLAS with the kernel chosen by MEASUREMENT -- rocBLAS' own
    choice for these shapes runs at 62-77 TFLOP/s on MI355X, its best candidate at ~100 (104/90/84 -> 65/62/60 us at B=32
    T=1000 S=200 C=500).  By default the library times the candidates at the second call with a shape (~0.2 s, never inside
    a stream capture), so a loop with fixed shapes gets the fast kernel from its second step on without calling anything
    and a ragged loop is never held up.  This function only moves that switch (the environment variable FTR_GEMM_TUNE):
    ``enable=False`` or ``search=False`` -> "off" (shapes already measured keep their kernel), ``when`` = "first" | "second".
    ``filename`` is accepted for compatibility and ignored (the choices live in the process; see
    ``normalizer_gemm_choice`` / ``set_normalizer_gemm_choice`` to carry one over)."""
    if when not in ("first", "second"):
        raise ValueError("when must be 'first' or 'second'")
    os.environ["FTR_GEMM_TUNE"] = when if (enable and search) else "off"


def normalizer_gemm_choice(kind: int, B: int, T: int, S: int, C: int):
    """What the library chose for the GEMM ``kind`` (0 forward product, 1 towards lm, 2 towards am) of this shape on the
    current device: None if the shape has not run, else dict(solution, us, us_default, candidates) -- solution 0 is
    rocBLAS' own choice, candidates -1 means not measured yet."""
    import ctypes
    sol, cand = ctypes.c_int(0), ctypes.c_int(0)
    us, usd = ctypes.c_float(0), ctypes.c_float(0)
    if not _lib.lib().ftr_normalizer_gemm_choice(int(kind), int(B), int(T), int(S) + 1, int(C), ctypes.byref(sol), ctypes.byref(us),
                                                  ctypes.byref(usd), ctypes.byref(cand)):
        return None
    return dict(solution=sol.value, us=us.value, us_default=usd.value, candidates=cand.value)


def set_normalizer_gemm_choice(kind: int, B: int, T: int, S: int, C: int, solution: int) -> None:
    """Fixes the rocBLAS solution index for a shape without measuring (a choice recorded by an earlier process)."""
    _lib.call("ftr_normalizer_gemm_set_choice", int(kind), int(B), int(T), int(S) + 1, int(C), int(solution))


def _gemm(kind: int, x: torch.Tensor, y: torch.Tensor, B: int, T: int, S: int, C: int, st) -> torch.Tensor:
    """ftr_normalizer_gemm_f32: kind 0 lm_probs . am_probs^T -> [B,S+1,T]; 1 W . am_probs -> [B,S+1,C]; 2 W^T . lm_probs -> [B,T,C]."""
    shape = ((B, S + 1, T), (B, S + 1, C), (B, T, C))[kind]
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _lib.call("ftr_normalizer_gemm_f32", kind, _ptr(x), _ptr(y), _ptr(out), B, T, S + 1, C, st)
    return out


def _check_type(rnnt_type: str) -> None:
    if rnnt_type not in ("regular", "modified", "constrained"):
        raise ValueError(f"rnnt_type should be ('regular' | 'modified' | 'constrained'), given {rnnt_type}")


def _i64(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int64 else t.to(torch.int64)


def fix_for_boundary(px: torch.Tensor, boundary: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rnnt_loss.py:28-61: px[b, :, boundary[b,3]] = -inf (regular type only)."""
    if boundary is None:
        return px
    B, S, T1 = px.shape
    idx = _i64(boundary[:, 3]).reshape(B, 1, 1).expand(B, S, 1)
    return px.scatter(2, idx, _NEG_INF)


def _use_fused_builder(C: int) -> bool:
    """The f32-MFMA builder kernel (csrc/simple_fused.hip) unless the size is outside its domain (C % 4 != 0) or
    FTR_BUILDER_GEMM=library asks for the library-GEMM route (A/B comparisons)."""
    return os.environ.get("FTR_BUILDER_GEMM", "fused") != "library" and bool(_lib.lib().ftr_simple_logprobs_fused_supported(int(C)))


def _use_fused_builder_bwd(T: int, C: int) -> bool:
    """The fused d am kernel (W^T lm_probs + the scatter by symbol inside one kernel, csrc/simple_fused.hip: `damp` [B,T,C]
    never goes through memory) where it is the faster route: measured on MI355X it beats library GEMM + epilogue kernel for
    large vocabularies (B=32 T=2000 S=300 C=1024: 644 us against 308 + 405 with the tuned GEMM) and loses at C = 500 (141 us
    against 63 + 62; 142 against the untuned GEMM), so the default ("auto") takes it from C >= 768.  FTR_BUILDER_BWD=fused /
    library force a route (A/B comparisons and the route-vs-route tests)."""
    mode = os.environ.get("FTR_BUILDER_BWD", "auto")
    if mode == "library" or not _use_fused_builder(C) or not _lib.lib().ftr_simple_logprobs_fused_bwd_supported(int(T), int(C)):
        return False
    return mode == "fused" or C >= 768


def _simple_builder(amc, lmc, symbols, am_probs, lm_probs, am_max, lm_max, boundary, blank, delay_penalty, px, py,
                    B, T, S, C, modified, st):
    """normalisers + px / py of get_rnnt_logprobs (rnnt_loss.py:180-221): one fused launch, or library GEMM + epilogue.
    Returns the product [B,S+1,T] (the backward's W kernel reads it)."""
    if _use_fused_builder(C):
        prod = torch.empty((B, S + 1, T), dtype=torch.float32, device=amc.device)
        _lib.call("ftr_simple_logprobs_fused_fwd_f32", _ptr(amc), _ptr(lmc), _ptr(symbols), _ptr(am_probs), _ptr(lm_probs),
                  _ptr(am_max), _ptr(lm_max), _ptr(boundary), int(blank), float(delay_penalty), _ptr(px), _ptr(py),
                  _ptr(prod), B, T, S, C, int(modified), st)
        return prod
    prod = _gemm(0, lm_probs, am_probs, B, T, S, C, st)                                               # :180-182
    _lib.call("ftr_simple_logprobs_fwd_f32", _ptr(amc), _ptr(lmc), _ptr(symbols), _ptr(prod), _ptr(am_max),
              _ptr(lm_max), _ptr(boundary), int(blank), float(delay_penalty), _ptr(px), _ptr(py),
              B, T, S, C, int(modified), st)
    return prod


class _SimpleLogprobs(torch.autograd.Function):
    """get_rnnt_logprobs (+ fix_for_boundary + delay penalty) for regular/modified: native prologue and
    epilogue kernels around the normaliser GEMM (ftr_normalizer_gemm_f32 -> rocBLAS), hand-written backward."""

    @staticmethod
    def forward(ctx, lm, am, symbols, termination_symbol, boundary, modified, delay_penalty):
        B, T, C = am.shape
        S = lm.shape[1] - 1
        T1 = T if modified else T + 1
        amc = am.detach().contiguous(); lmc = lm.detach().contiguous()
        dev = amc.device
        am_probs = torch.empty_like(amc); lm_probs = torch.empty_like(lmc)
        am_max = torch.empty((B, T), dtype=torch.float32, device=dev)
        lm_max = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        px = torch.empty((B, S, T1), dtype=torch.float32, device=dev)
        py = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream_ptr(amc)
            _lib.call("ftr_rowmax_exp_pair_f32", _ptr(amc), _ptr(am_probs), _ptr(am_max), B * T,              # :175-178
                      _ptr(lmc), _ptr(lm_probs), _ptr(lm_max), B * (S + 1), C, st)
            prod = _simple_builder(amc, lmc, symbols, am_probs, lm_probs, am_max, lm_max, boundary, termination_symbol,
                                   delay_penalty, px, py, B, T, S, C, modified, st)
        ctx.save_for_backward(am_probs, lm_probs, prod, symbols, boundary if boundary is not None else torch.empty(0))
        ctx.has_boundary = boundary is not None
        ctx.meta = (int(termination_symbol), int(modified))
        return px, py

    @staticmethod
    def backward(ctx, gpx, gpy):
        am_probs, lm_probs, prod, symbols, boundary = ctx.saved_tensors
        if not ctx.has_boundary:
            boundary = None
        blank, modified = ctx.meta
        B, T, C = am_probs.shape
        S = lm_probs.shape[1] - 1
        dev = am_probs.device
        gpx = gpx.contiguous(); gpy = gpy.contiguous()
        W = torch.empty_like(prod)
        rsx = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        rsy = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        d_am = torch.empty_like(am_probs); d_lm = torch.empty_like(lm_probs)
        with torch.cuda.device(dev):
            st = _stream_ptr(am_probs)
            _lib.call("ftr_simple_logprobs_bwd_w_f32", _ptr(gpx), _ptr(gpy), _ptr(prod), _ptr(boundary), _ptr(W),
                      _ptr(rsx), _ptr(rsy), B, T, S, modified, st)
            dlmp = _gemm(1, W, am_probs, B, T, S, C, st)         # [B,S+1,C]
            if _use_fused_builder_bwd(T, C):                       # W^T lm_probs inside the d am kernel (opt-in)
                _lib.call("ftr_simple_logprobs_fused_bwd_am_f32", _ptr(gpx), _ptr(gpy), None, 0, 1.0, _ptr(prod),
                          _ptr(lm_probs), _ptr(am_probs), _ptr(symbols), _ptr(boundary), blank, _ptr(d_am), B, T, S, C,
                          modified, st)
            else:
                damp = _gemm(2, W, lm_probs, B, T, S, C, st)    # [B,T,C]
                _lib.call("ftr_simple_logprobs_bwd_am_f32", _ptr(gpx), _ptr(gpy), _ptr(damp), _ptr(am_probs),
                          _ptr(symbols), _ptr(boundary), blank, _ptr(d_am), B, T, S, C, modified, st)
            _lib.call("ftr_simple_logprobs_bwd_lm_f32", _ptr(dlmp), _ptr(lm_probs), _ptr(symbols), _ptr(rsx), _ptr(rsy),
                      blank, _ptr(d_lm), B, S, C, st)
        return d_lm, d_am, None, None, None, None, None


class _SimpleLoss(torch.autograd.Function):
    """rnnt_loss_simple for regular/modified as ONE graph node: builder kernels + GEMM, recursion forward + backward
    (occupancies), native loss reduction; backward() feeds the occupancies with the upstream gradient folded in on
    the fly (ftr_simple_logprobs_bwd_*_scaled_f32) -- no framework-side pass over a lattice anywhere."""

    @staticmethod
    def forward(ctx, lm, am, symbols, termination_symbol, boundary, modified, delay_penalty, code, want_occupancies):
        B, T, C = am.shape
        S = lm.shape[1] - 1
        T1 = T if modified else T + 1
        amc = am.detach().contiguous(); lmc = lm.detach().contiguous()
        dev = amc.device
        am_probs = torch.empty_like(amc); lm_probs = torch.empty_like(lmc)
        am_max = torch.empty((B, T), dtype=torch.float32, device=dev)
        lm_max = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        px = torch.empty((B, S, T1), dtype=torch.float32, device=dev)
        py = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream_ptr(amc)
            _lib.call("ftr_rowmax_exp_pair_f32", _ptr(amc), _ptr(am_probs), _ptr(am_max), B * T,              # :175-178
                      _ptr(lmc), _ptr(lm_probs), _ptr(lm_max), B * (S + 1), C, st)
            prod = _simple_builder(amc, lmc, symbols, am_probs, lm_probs, am_max, lm_max, boundary, termination_symbol,
                                   delay_penalty, px, py, B, T, S, C, modified, st)
        # the recursion backward (occupancies) only when somebody wants them: the caller (calc_gradients) or autograd
        need = bool(want_occupancies) or ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if need:      # the loss tail rides along with the recursion's backward launch
            ans, px_grad, py_grad, loss = mi_forward_backward(px, py, boundary, True, ans_grad_is_one=True, loss_code=code)
        else:
            ans, px_grad, py_grad = mi_forward_backward(px, py, boundary, False, ans_grad_is_one=True)
            loss = _negated_reduce_native(ans, code)
        del px, py
        if need:
            ctx.save_for_backward(am_probs, lm_probs, prod, symbols, boundary if boundary is not None else torch.empty(0),
                                  px_grad, py_grad)
        else:
            px_grad = torch.zeros((B, S, T1), dtype=torch.float32, device=dev)
            py_grad = torch.zeros((B, S + 1, T), dtype=torch.float32, device=dev)
        ctx.has_boundary = boundary is not None
        ctx.meta = (int(termination_symbol), int(modified), int(code))
        ctx.mark_non_differentiable(px_grad, py_grad)
        ctx.set_materialize_grads(False)          # no zero tensors for the two occupancy outputs in backward
        return loss, px_grad, py_grad

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        if g_loss is None:
            return (None,) * 9
        am_probs, lm_probs, prod, symbols, boundary, px_grad, py_grad = ctx.saved_tensors
        if not ctx.has_boundary:
            boundary = None
        blank, modified, code = ctx.meta
        B, T, C = am_probs.shape
        S = lm_probs.shape[1] - 1
        dev = am_probs.device
        scale, stride, mul = _upstream_scale(g_loss, code, B)
        W = torch.empty_like(prod)
        rsx = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        rsy = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        d_am = torch.empty_like(am_probs); d_lm = torch.empty_like(lm_probs)
        with torch.cuda.device(dev):
            st = _stream_ptr(am_probs)
            _lib.call("ftr_simple_logprobs_bwd_w_scaled_f32", _ptr(px_grad), _ptr(py_grad), _ptr(scale), stride, mul,
                      _ptr(prod), _ptr(boundary), _ptr(W), _ptr(rsx), _ptr(rsy), B, T, S, modified, st)
            dlmp = _gemm(1, W, am_probs, B, T, S, C, st)         # [B,S+1,C]
            if _use_fused_builder_bwd(T, C):                       # W^T lm_probs inside the d am kernel (opt-in)
                _lib.call("ftr_simple_logprobs_fused_bwd_am_f32", _ptr(px_grad), _ptr(py_grad), _ptr(scale), stride, mul,
                          _ptr(prod), _ptr(lm_probs), _ptr(am_probs), _ptr(symbols), _ptr(boundary), blank, _ptr(d_am),
                          B, T, S, C, modified, st)
            else:
                damp = _gemm(2, W, lm_probs, B, T, S, C, st)    # [B,T,C]
                _lib.call("ftr_simple_logprobs_bwd_am_scaled_f32", _ptr(px_grad), _ptr(py_grad), _ptr(scale), stride, mul,
                          _ptr(damp), _ptr(am_probs), _ptr(symbols), _ptr(boundary), blank, _ptr(d_am), B, T, S, C,
                          modified, st)
            _lib.call("ftr_simple_logprobs_bwd_lm_f32", _ptr(dlmp), _ptr(lm_probs), _ptr(symbols), _ptr(rsx), _ptr(rsy),
                      blank, _ptr(d_lm), B, S, C, st)
        return d_lm, d_am, None, None, None, None, None, None, None


def _check_simple_inputs(lm, am, symbols, termination_symbol):
    _require_gpu(am, "am"); _require_gpu(lm, "lm")
    if am.dtype != torch.float32 or lm.dtype != torch.float32:
        raise TypeError("am and lm must be float32")
    B, T, C = am.shape
    S = lm.shape[1] - 1
    if lm.shape[0] != B or lm.shape[2] != C:
        raise ValueError(f"lm {tuple(lm.shape)} and am {tuple(am.shape)} disagree")
    symbols = torch.as_tensor(symbols, device=am.device)
    if tuple(symbols.shape) != (B, S):
        raise ValueError(f"symbols must have shape {(B, S)}, got {tuple(symbols.shape)}")
    if not 0 <= int(termination_symbol) < C:
        raise ValueError(f"termination_symbol {termination_symbol} not in [0, {C})")
    return symbols.to(torch.int32).contiguous()


def _simple_logprobs_native(lm, am, symbols, termination_symbol, rnnt_type, boundary, delay_penalty=0.0):
    _require_gpu(am, "am"); _require_gpu(lm, "lm")
    if am.dtype != torch.float32 or lm.dtype != torch.float32:
        raise TypeError("am and lm must be float32")
    B, T, C = am.shape
    S = lm.shape[1] - 1
    if lm.shape[0] != B or lm.shape[2] != C:
        raise ValueError(f"lm {tuple(lm.shape)} and am {tuple(am.shape)} disagree")
    symbols = torch.as_tensor(symbols, device=am.device)
    if tuple(symbols.shape) != (B, S):
        raise ValueError(f"symbols must have shape {(B, S)}, got {tuple(symbols.shape)}")
    if not 0 <= int(termination_symbol) < C:
        raise ValueError(f"termination_symbol {termination_symbol} not in [0, {C})")
    symbols = symbols.to(torch.int32).contiguous()
    boundary = _as_boundary(boundary, B, am.device)
    modified = rnnt_type != "regular"
    pen = float(delay_penalty) if delay_penalty > 0.0 else 0.0
    px, py = _SimpleLogprobs.apply(lm, am, symbols, termination_symbol, boundary, modified, pen)
    if rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return px, py


def get_rnnt_logprobs(
    lm: torch.Tensor,
    am: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    rnnt_type: str = "regular",
    boundary: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """rnnt_loss.py:63-223.  lm [B,S+1,C], am [B,T,C], symbols [B,S] -> px [B,S,T+1|T], py [B,S+1,T].
    Native (HIP) prologue/epilogue around one library GEMM; differentiable w.r.t. lm and am."""
    _check_type(rnnt_type)
    return _simple_logprobs_native(lm, am, symbols, termination_symbol, rnnt_type, boundary)


def _apply_delay_penalty(px, boundary, rnnt_type, delay_penalty):
    """rnnt_loss.py:305-321 (also :518-534, :1097-1114, :1461-1478): float64 offsets, cast to px.dtype."""
    if not delay_penalty > 0.0:
        return px
    B, S, T0 = px.shape
    T = T0 if rnnt_type != "regular" else T0 - 1
    if boundary is None:
        offset = torch.full((B,), (T - 1) / 2, dtype=torch.float64, device=px.device)
    else:
        offset = (boundary[:, 3].to(torch.float64) - 1) / 2
    penalty = offset.reshape(B, 1, 1) - torch.arange(T0, dtype=torch.float64, device=px.device).reshape(1, 1, T0)
    penalty = penalty * delay_penalty
    return px + penalty.to(px.dtype)


def _reduce(negated_loss: torch.Tensor, reduction: Optional[str]) -> torch.Tensor:
    if reduction == "none":
        return -negated_loss
    if reduction == "mean":
        return -torch.mean(negated_loss)
    if reduction == "sum":
        return -torch.sum(negated_loss)
    raise ValueError(f"reduction should be ('none' | 'mean' | 'sum'), given {reduction}")


_REDUCTIONS = {"none": 0, "mean": 1, "sum": 2}


def _reduction_code(reduction: Optional[str]) -> int:
    if reduction not in _REDUCTIONS:
        raise ValueError(f"reduction should be ('none' | 'mean' | 'sum'), given {reduction}")
    return _REDUCTIONS[reduction]


def _negated_reduce_native(ans: torch.Tensor, code: int) -> torch.Tensor:
    """-ans / -mean / -sum in one native launch (the loss tail of rnnt_loss.py:333,544-546,1124-1126,1487-1489)."""
    B = ans.shape[0]
    out = torch.empty((B,) if code == 0 else (), dtype=torch.float32, device=ans.device)
    with torch.cuda.device(ans.device):
        _lib.call("ftr_negated_reduce_f32", _ptr(ans), B, code, _ptr(out), _stream_ptr(ans))
    return out


def _upstream_scale(g_loss: torch.Tensor, code: int, B: int):
    """(pointer tensor, stride, multiplier) such that d loss / d ans[b] = scale[b * stride] * mul."""
    g = g_loss.to(torch.float32).contiguous()
    return g, (1 if code == 0 else 0), (-1.0 / B if code == 1 else -1.0)


def _drive(px, py, boundary, reduction, calc_gradients):
    scores_and_grads = mutual_information_recursion(px=px, py=py, boundary=boundary, calc_gradients=calc_gradients)
    negated_loss = scores_and_grads[0] if calc_gradients else scores_and_grads
    loss = _reduce(negated_loss, reduction)
    return (loss, scores_and_grads[1]) if calc_gradients else loss


def rnnt_loss_simple(
    lm: torch.Tensor,
    am: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    boundary: Optional[torch.Tensor] = None,
    rnnt_type: str = "regular",
    delay_penalty: float = 0.0,
    reduction: Optional[str] = "mean",
    calc_gradients: bool = False,
) -> Union[torch.Tensor, Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]]:
    """rnnt_loss.py:225-338.  Returns loss, or (loss, (px_grad, py_grad)) when ``calc_gradients``."""
    _check_type(rnnt_type)
    code = _reduction_code(reduction)
    boundary = _as_boundary(boundary, am.shape[0], am.device)
    if rnnt_type == "constrained":   # the penalty applies after px += py[:, 1:, :]  (:218-221, :305-321)
        px, py = _simple_logprobs_native(lm, am, symbols, termination_symbol, rnnt_type, boundary)
        px = _apply_delay_penalty(px, boundary, rnnt_type, delay_penalty)
        return _drive(px, py, boundary, reduction, calc_gradients)
    symbols_i = _check_simple_inputs(lm, am, symbols, termination_symbol)
    pen = float(delay_penalty) if delay_penalty > 0.0 else 0.0
    loss, px_grad, py_grad = _SimpleLoss.apply(lm, am, symbols_i, termination_symbol, boundary, rnnt_type != "regular",
                                               pen, code, bool(calc_gradients))
    return (loss, (px_grad, py_grad)) if calc_gradients else loss


def _identity_ranges(B: int, T: int, S1: int, device) -> torch.Tensor:
    """ranges[b,t,:] = 0..S: with every row in range the pruned builder IS the joint builder (its band is the lattice)."""
    ranges = torch.arange(S1, dtype=torch.int32, device=device).expand(B, T, S1).contiguous()
    _mark_band(ranges)
    return ranges


def get_rnnt_logprobs_joint(
    logits: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    boundary: Optional[torch.Tensor] = None,
    rnnt_type: str = "regular",
) -> Tuple[torch.Tensor, torch.Tensor]:
    """rnnt_loss.py:340-452.  logits [B,T,S+1,C] -> px [B,S,T+1|T], py [B,S+1,T].  Native: the streaming
    log-sum-exp + lattice writer of the pruned builder with s_range = S+1 (identity ranges), and the same
    hand-written backward (d/d logits = scattered gradient - softmax * row sum)."""
    _check_type(rnnt_type)
    if logits.dim() != 4:
        raise ValueError("logits must be [B,T,S+1,C]")
    B, T, S1, C = logits.shape
    if tuple(torch.as_tensor(symbols).shape) != (B, S1 - 1):
        raise ValueError(f"symbols must have shape {(B, S1 - 1)}, got {tuple(torch.as_tensor(symbols).shape)}")
    _require_gpu(logits, "logits")
    return get_rnnt_logprobs_pruned(logits=logits, symbols=symbols, ranges=_identity_ranges(B, T, S1, logits.device),
                                    termination_symbol=termination_symbol, boundary=boundary, rnnt_type=rnnt_type)


def rnnt_loss(
    logits: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    boundary: Optional[torch.Tensor] = None,
    rnnt_type: str = "regular",
    delay_penalty: float = 0.0,
    reduction: Optional[str] = "mean",
    calc_gradients: bool = False,
) -> torch.Tensor:
    """rnnt_loss.py:454-551 (unpruned loss on joiner logits [B,T,S+1,C]).  Without ``calc_gradients`` this is the
    fused pruned loss with identity ranges (px/py, recursion forward+backward and d/d logits in native code)."""
    _check_type(rnnt_type)
    boundary = _as_boundary(boundary, logits.shape[0], logits.device)
    if not calc_gradients:
        _require_gpu(logits, "logits")
        B, T, S1, _ = logits.shape
        return rnnt_loss_pruned(logits=logits, symbols=symbols, ranges=_identity_ranges(B, T, S1, logits.device),
                                termination_symbol=termination_symbol, boundary=boundary, rnnt_type=rnnt_type,
                                delay_penalty=delay_penalty, reduction=reduction)
    px, py = get_rnnt_logprobs_joint(logits=logits, symbols=symbols, termination_symbol=termination_symbol,
                                     boundary=boundary, rnnt_type=rnnt_type)
    px = _apply_delay_penalty(px, boundary, rnnt_type, delay_penalty)
    return _drive(px, py, boundary, reduction, calc_gradients)


def _monotonic_lower_bound(x: torch.Tensor) -> torch.Tensor:
    """rnnt_loss.py:553-585: reverse -> cummin -> reverse (suffix minimum).  int32, last axis."""
    squeeze = x.dim() == 1
    x2 = x.reshape(1, -1) if squeeze else x
    out = cummin(torch.flip(x2.to(torch.int32), dims=(-1,)).contiguous())
    out = torch.flip(out, dims=(-1,))
    return out[0] if squeeze else out


def _adjust_pruning_lower_bound(s_begin: torch.Tensor, s_range: int) -> torch.Tensor:
    """rnnt_loss.py:587-641, op by op on top of the native ``cummin`` (the fused path used by
    ``get_rnnt_prune_ranges`` does the same arithmetic in one kernel)."""
    B, T = s_begin.shape
    ar = torch.arange(0, T, dtype=torch.int32, device=s_begin.device)
    s_begin = _monotonic_lower_bound(s_begin)
    s_begin = -(s_begin - (s_range - 1) * ar)
    s_begin = _monotonic_lower_bound(s_begin)
    s_begin = torch.clamp(s_begin, min=0)
    s_begin = -(s_begin - (s_range - 1) * ar)
    return s_begin


def get_rnnt_prune_ranges(
    px_grad: torch.Tensor,
    py_grad: torch.Tensor,
    boundary: torch.Tensor,
    s_range: int,
) -> torch.Tensor:
    """rnnt_loss.py:647-761.  Returns int32 ranges [B,T,s_range'] with s_range' = S+1 if s_range > S."""
    _require_gpu(px_grad, "px_grad"); _require_gpu(py_grad, "py_grad")
    B, S, T1 = px_grad.shape
    T = py_grad.shape[-1]
    if T1 not in (T, T + 1):
        raise ValueError(f"px_grad.shape[-1]={T1} must be T or T+1 (T={T})")
    if tuple(py_grad.shape) != (B, S + 1, T):
        raise ValueError(f"py_grad must have shape {(B, S + 1, T)}, got {tuple(py_grad.shape)}")
    if boundary is None:
        raise ValueError("get_rnnt_prune_ranges: boundary is mandatory (rnnt_loss.py:741-746)")
    s_range = int(s_range)
    r = S + 1 if s_range > S else s_range
    px_grad = px_grad.detach().to(torch.float32).contiguous()
    py_grad = py_grad.detach().to(torch.float32).contiguous()
    boundary = _as_boundary(boundary, B, px_grad.device)
    ranges = torch.empty((B, T, r), dtype=torch.int32, device=px_grad.device)
    scratch = torch.empty((B, T), dtype=torch.int32, device=px_grad.device)
    import ctypes
    r_eff = ctypes.c_int(0)
    with torch.cuda.device(px_grad.device):
        _lib.call("ftr_prune_ranges_i32", _ptr(px_grad), _ptr(py_grad), _ptr(boundary), _ptr(ranges),
                                                   _ptr(scratch), B, S, T, T1, s_range, ctypes.byref(r_eff),
                                                   _stream_ptr(px_grad))
    assert r_eff.value == r
    # what the kernel guarantees (rnnt_loss.py:673-677): ranges[b,t,0] non-decreasing in t, inside [0, S - r + 1], and
    # ranges[b,t,k] = ranges[b,t,0] + k: a band.  The mark only saves rnnt_loss_pruned the device-side check it runs on
    # ranges tensors it has not seen (a clone, a slice, a reloaded tensor: _is_band).
    _mark_band(ranges)
    return ranges


class _DoPruning(torch.autograd.Function):
    @staticmethod
    def forward(ctx, am, lm, ranges, dense):
        B, T, r = ranges.shape
        S1, C = lm.shape[1], lm.shape[2]
        am_c = am.detach().contiguous(); lm_c = lm.detach().contiguous()
        # am_pruned[b,t,k,:] = am[b,t,:] for every k (rnnt_loss.py:803): a broadcast.  TensorFlow has no strided tensors and
        # materialises it; here it stays a stride-0 view (as k2's fast_rnnt returns it), which a joiner's `am_pruned +
        # lm_pruned` consumes by broadcasting: B*T*r*C*4 bytes less to write here and to read there.  Values, shape and the
        # gradient (the sum over r, in the fused backward below) are the reference's; .contiguous() gives the dense tensor.
        # dense=True materialises it (one more [B,T,r,C] stream written by the same kernel): for callers that write into
        # am_pruned, reshape it with .view(), or hand it to code that wants contiguous memory.
        am_p = torch.empty((B, T, r, C), dtype=am.dtype, device=am.device) if dense else am_c.unsqueeze(2).expand(B, T, r, C)
        lm_p = torch.empty((B, T, r, C), dtype=lm.dtype, device=lm.device)
        with torch.cuda.device(am.device):
            _lib.call("ftr_do_pruning_f32", _ptr(am_c), _ptr(lm_c), _ptr(ranges), _ptr(am_p) if dense else None, _ptr(lm_p),
                                                     B, T, S1, C, r, _stream_ptr(am))
        ctx.save_for_backward(ranges)
        ctx.lm_shape = tuple(lm.shape)
        return am_p, lm_p

    @staticmethod
    def backward(ctx, g_am_p, g_lm_p):
        (ranges,) = ctx.saved_tensors
        B, S1, C = ctx.lm_shape
        T, r = ranges.shape[1], ranges.shape[2]
        g_am_p = g_am_p.contiguous(); g_lm_p = g_lm_p.contiguous()
        g_am = torch.empty((B, T, C), dtype=g_am_p.dtype, device=g_am_p.device)     # broadcast <-> sum over s_range
        g_lm = torch.empty((B, S1, C), dtype=g_lm_p.dtype, device=g_lm_p.device)    # gather    <-> segment sum
        ws_bytes = int(_lib.lib().ftr_do_pruning_bwd_workspace_bytes(B, T, S1, C, r))
        ws = torch.empty(((ws_bytes + 3) // 4,), dtype=torch.float32, device=g_am_p.device) if ws_bytes else None
        with torch.cuda.device(g_am_p.device):
            _lib.call("ftr_do_pruning_bwd_ws_f32", _ptr(g_am_p), _ptr(g_lm_p), _ptr(ranges), _ptr(g_am), _ptr(g_lm),
                      B, T, S1, C, r, _ptr(ws), ws_bytes, _stream_ptr(g_am_p))
        return g_am, g_lm, None, None


def do_rnnt_pruning(am: torch.Tensor, lm: torch.Tensor, ranges: torch.Tensor, dense: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """rnnt_loss.py:763-812.  am [B,T,C], lm [B,S+1,C], ranges [B,T,s_range] -> two [B,T,s_range,C].

    ``am_pruned[b,t,k,:] = am[b,t,:]`` is a broadcast (rnnt_loss.py:803).  By default it is returned as a stride-0 VIEW of
    ``am`` (as k2's fast_rnnt returns it): same values, shape and gradient as the reference's dense tensor, and a joiner's
    ``am_pruned + lm_pruned`` reads it by broadcasting -- but it aliases ``am`` (changing ``am`` afterwards changes it),
    it cannot be written in place and ``.view(-1, C)`` refuses it.  ``dense=True`` (an extension, not in the reference's
    signature) returns the reference's materialised tensor, written by the same gather kernel; ``.contiguous()`` on the
    view gives the same thing."""
    _require_gpu(am, "am"); _require_gpu(lm, "lm"); _require_gpu(ranges, "ranges")
    if am.dtype != torch.float32 or lm.dtype != torch.float32:
        raise TypeError("am and lm must be float32")
    if ranges.shape[0] != am.shape[0] or ranges.shape[0] != lm.shape[0] or am.shape[1] != ranges.shape[1]:
        raise ValueError("do_rnnt_pruning: inconsistent shapes")
    ranges = ranges.to(torch.int32).contiguous()
    return _DoPruning.apply(am, lm, ranges, bool(dense))


class _PrunedLogprobs(torch.autograd.Function):
    """get_rnnt_logprobs_pruned for regular/modified as two native launches each way."""

    @staticmethod
    def forward(ctx, logits, symbols, ranges, termination_symbol, boundary, modified, delay_penalty):
        B, T, r, C = logits.shape
        S = symbols.shape[1]
        T1 = T if modified else T + 1
        x = logits.detach().contiguous()
        lse = torch.empty((B, T, r), dtype=torch.float32, device=x.device)
        px = torch.empty((B, S, T1), dtype=torch.float32, device=x.device)
        py = torch.empty((B, S + 1, T), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("ftr_pruned_logprobs_fwd_f32", _ptr(x), _ptr(symbols), _ptr(ranges), _ptr(boundary),
                                                              int(termination_symbol), float(delay_penalty), _ptr(lse),
                                                              _ptr(px), _ptr(py), B, T, S, C, r, int(modified),
                                                              _stream_ptr(x))
        ctx.save_for_backward(x, symbols, ranges, lse, boundary if boundary is not None else torch.empty(0))
        ctx.has_boundary = boundary is not None
        ctx.meta = (int(termination_symbol), int(modified))
        return px, py

    @staticmethod
    def backward(ctx, gpx, gpy):
        x, symbols, ranges, lse, boundary = ctx.saved_tensors
        if not ctx.has_boundary:
            boundary = None
        blank, modified = ctx.meta
        B, T, r, C = x.shape
        S = symbols.shape[1]
        g = torch.empty_like(x)
        gpx = gpx.contiguous(); gpy = gpy.contiguous()
        with torch.cuda.device(x.device):
            _lib.call("ftr_pruned_logprobs_bwd_f32", _ptr(x), _ptr(symbols), _ptr(ranges), _ptr(boundary), blank,
                                                              _ptr(lse), _ptr(gpx), _ptr(gpy), None, _ptr(g),
                                                              B, T, S, C, r, modified, _stream_ptr(x))
        return g, None, None, None, None, None, None


def _pruned_inputs(logits, symbols, ranges, boundary):
    _require_gpu(logits, "logits")
    if logits.dim() != 4:
        raise ValueError("logits must be [B,T,s_range,C]")
    if logits.dtype != torch.float32:
        raise TypeError("logits must be float32")
    B, T, r, C = logits.shape
    symbols = torch.as_tensor(symbols, device=logits.device).to(torch.int32).contiguous()
    ranges_in = ranges
    ranges = torch.as_tensor(ranges, device=logits.device).to(torch.int32).contiguous()
    if ranges is not ranges_in and isinstance(ranges_in, torch.Tensor) and _band_mark_valid(ranges_in):
        _mark_band(ranges)      # a dtype / layout / device copy of a known band is still a band
    if tuple(ranges.shape) != (B, T, r):
        raise ValueError(f"ranges must have shape {(B, T, r)}, got {tuple(ranges.shape)}")
    if symbols.dim() != 2 or symbols.shape[0] != B:
        raise ValueError("symbols must be [B,S]")
    boundary = _as_boundary(boundary, B, logits.device)
    return symbols, ranges, boundary


def get_rnnt_logprobs_pruned(
    logits: torch.Tensor,
    symbols: torch.Tensor,
    ranges: torch.Tensor,
    termination_symbol: int,
    boundary: torch.Tensor,
    rnnt_type: str = "regular",
) -> Tuple[torch.Tensor, torch.Tensor]:
    """rnnt_loss.py:853-1020.  logits [B,T,s_range,C] -> full-size px [B,S,T+1|T], py [B,S+1,T] with
    -inf outside the pruned band."""
    _check_type(rnnt_type)
    symbols, ranges, boundary = _pruned_inputs(logits, symbols, ranges, boundary)
    modified = rnnt_type != "regular"
    px, py = _PrunedLogprobs.apply(logits, symbols, ranges, termination_symbol, boundary, modified, 0.0)
    if rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return px, py


def _mark_band(ranges: torch.Tensor) -> None:
    """Remembers on the tensor object that its CURRENT contents are a band (the version counter catches in-place edits)."""
    ranges._ftr_band = ranges._version


def _band_mark_valid(ranges: torch.Tensor) -> bool:
    return getattr(ranges, "_ftr_band", None) == ranges._version


def _is_band(ranges: torch.Tensor, boundary) -> bool:
    """Are these ranges what the band-native kernels assume (ranges[b,t,0] non-decreasing over the frames of each boundary
    rectangle, ranges[b,t,k] = ranges[b,t,0] + k)?  Decided by the DATA: a tensor that carries no valid mark (anything but
    the output of get_rnnt_prune_ranges itself: a clone, a slice, a .to(), a checkpoint) is checked once on the device
    (ftr_band_ranges_check_i32, one small kernel and one host read of its flag word) and the verdict is remembered on the
    tensor, so only the first use of an unknown tensor synchronises.  Under stream capture an unknown tensor takes the
    lattice route (no host read inside a capture)."""
    if _band_mark_valid(ranges):
        return True
    if getattr(ranges, "_ftr_not_band", None) == ranges._version:
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    B, T, r = ranges.shape
    flags = torch.empty((1,), dtype=torch.int32, device=ranges.device)
    with torch.cuda.device(ranges.device):
        _lib.call("ftr_band_ranges_check_i32", _ptr(ranges), _ptr(boundary), _ptr(flags), B, T, r, _stream_ptr(ranges))
    ok = int(flags.item()) == 0
    if ok:
        _mark_band(ranges)
    else:
        ranges._ftr_not_band = ranges._version
    return ok


def _band_path_ok(ranges, boundary, T: int, S: int, r: int) -> bool:
    """The band-native recursion needs a band that its kernels cover (r <= 15) and ranges that ARE a band (_is_band).
    FTR_PRUNED_ROUTE=lattice (a test knob, read at call time) sends everything through the full-size lattices."""
    import os
    if os.environ.get("FTR_PRUNED_ROUTE") == "lattice":
        return False
    return _lib.lib().ftr_mutual_information_band_supported(int(T), int(S), int(r)) != 0 and _is_band(ranges, boundary)


class _PrunedLoss(torch.autograd.Function):
    """rnnt_loss_pruned for regular/modified with the whole chain native.

    Band path (ranges from get_rnnt_prune_ranges, band fits the kernel): logsumexp + band gather -> forward recursion,
    cut and backward recursion on the band [B,T,r] in one launch (ftr_mutual_information_band_ws_f32; no full-size lattice
    exists) -> in backward() one streaming kernel turns the band-shaped occupancies * upstream gradient into
    d loss / d logits.
    Lattice path (any other ranges): logsumexp + band->lattice, recursion forward / backward on the full-size lattices
    (what the reference does, rnnt_loss.py:968-1013), the same streaming kernel fed from the lattices."""

    @staticmethod
    def forward(ctx, logits, symbols, ranges, termination_symbol, boundary, modified, delay_penalty, code):
        B, T, r, C = logits.shape
        S = symbols.shape[1]
        T1 = T if modified else T + 1
        x = logits.detach().contiguous()
        need = logits.requires_grad
        lse = torch.empty((B, T, r), dtype=torch.float32, device=x.device)
        ctx.band = _band_path_ok(ranges, boundary, T, S, r)
        if ctx.band:
            pxb = torch.empty((B, T, r), dtype=torch.float32, device=x.device)
            pyb = torch.empty((B, T, r), dtype=torch.float32, device=x.device)
            gxb = torch.empty((B, T, r), dtype=torch.float32, device=x.device)
            gyb = torch.empty((B, T, r), dtype=torch.float32, device=x.device)
            ans = torch.empty((B,), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                st = _stream_ptr(x)
                _lib.call("ftr_pruned_band_fwd_f32", _ptr(x), _ptr(symbols), _ptr(ranges), _ptr(boundary),
                          int(termination_symbol), float(delay_penalty), _ptr(lse), _ptr(pxb), _ptr(pyb),
                          B, T, S, C, r, int(modified), st)
                nws = _lib.lib().ftr_mutual_information_band_workspace_floats(B, T, S, r)   # 0: the LDS-resident kernel
                bws = torch.empty((nws,), dtype=torch.float32, device=x.device) if nws else None
                _lib.call("ftr_mutual_information_band_ws_f32", _ptr(pxb), _ptr(pyb), _ptr(ranges), _ptr(boundary), _ptr(bws),
                          nws, _ptr(ans), _ptr(gxb), _ptr(gyb), B, T, S, r, int(modified), st)
                del bws
            del pxb, pyb
            if need:
                ctx.save_for_backward(x, symbols, ranges, lse, gxb, gyb,
                                      boundary if boundary is not None else torch.empty(0))
        else:
            px = torch.empty((B, S, T1), dtype=torch.float32, device=x.device)
            py = torch.empty((B, S + 1, T), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.call("ftr_pruned_logprobs_fwd_f32", _ptr(x), _ptr(symbols), _ptr(ranges), _ptr(boundary),
                                                                  int(termination_symbol), float(delay_penalty), _ptr(lse),
                                                                  _ptr(px), _ptr(py), B, T, S, C, r, int(modified),
                                                                  _stream_ptr(x))
            ans, px_grad, py_grad = mi_forward_backward(px, py, boundary, need, ans_grad_is_one=True)
            del px, py
            if need:
                ctx.save_for_backward(x, symbols, ranges, lse, px_grad, py_grad,
                                      boundary if boundary is not None else torch.empty(0))
        ctx.has_boundary = boundary is not None
        ctx.meta = (int(termination_symbol), int(modified), int(code))
        return _negated_reduce_native(ans, code)

    @staticmethod
    def backward(ctx, g_loss):
        x, symbols, ranges, lse, px_grad, py_grad, boundary = ctx.saved_tensors
        if not ctx.has_boundary:
            boundary = None
        blank, modified, code = ctx.meta
        B, T, r, C = x.shape
        S = symbols.shape[1]
        g = torch.empty_like(x)
        scale, stride, mul = _upstream_scale(g_loss, code, B)
        name = "ftr_pruned_band_bwd_scaled_f32" if ctx.band else "ftr_pruned_logprobs_bwd_scaled_f32"
        with torch.cuda.device(x.device):
            _lib.call(name, _ptr(x), _ptr(symbols), _ptr(ranges), _ptr(boundary), blank,
                      _ptr(lse), _ptr(px_grad), _ptr(py_grad), _ptr(scale), stride, mul, _ptr(g),
                      B, T, S, C, r, modified, _stream_ptr(x))
        return g, None, None, None, None, None, None, None


def rnnt_loss_pruned(
    logits: torch.Tensor,
    symbols: torch.Tensor,
    ranges: torch.Tensor,
    termination_symbol: int,
    boundary: torch.Tensor = None,
    rnnt_type: str = "regular",
    delay_penalty: float = 0.0,
    reduction: Optional[str] = "mean",
    calc_gradients: bool = False,
) -> torch.Tensor:
    """rnnt_loss.py:1022-1130.  Returns the loss only (``calc_gradients`` is accepted and, as in the
    reference, only selects whether the op computes occupancies; here that follows ``requires_grad``)."""
    _check_type(rnnt_type)
    code = _reduction_code(reduction)
    symbols_i, ranges_i, boundary_i = _pruned_inputs(logits, symbols, ranges, boundary)
    if rnnt_type == "constrained":
        px, py = get_rnnt_logprobs_pruned(logits=logits, symbols=symbols_i, ranges=ranges_i,
                                          termination_symbol=termination_symbol, boundary=boundary_i, rnnt_type=rnnt_type)
        px = _apply_delay_penalty(px, boundary_i, rnnt_type, delay_penalty)
        negated_loss = mutual_information_recursion(px=px, py=py, boundary=boundary_i, calc_gradients=False)
        return _reduce(negated_loss, reduction)
    return _PrunedLoss.apply(logits, symbols_i, ranges_i, termination_symbol, boundary_i, rnnt_type != "regular",
                             float(delay_penalty) if delay_penalty > 0.0 else 0.0, code)


def _colsum_weighted(x: torch.Tensor, w: torch.Tensor, rows: int, C: int, st) -> torch.Tensor:
    """out[c] = sum_row w[row] * x[row, c] on the native two-stage kernel (deterministic)."""
    n = _lib.lib().ftr_colsum_weighted_workspace_floats(rows, C)
    ws = torch.empty((max(n, 1),), dtype=torch.float32, device=x.device)
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    _lib.call("ftr_colsum_weighted_f32", _ptr(x), _ptr(w), _ptr(out), _ptr(ws), n, rows, C, st)
    return out


def _smoothed_forward(lm, am, symbols, termination_symbol, boundary, modified, lm_only_scale, am_only_scale,
                      process_group, delay_penalty):
    """Forward of the smoothed builder on the native kernels (rnnt_loss.py:1265-1365; with the penalty block :1461-1478
    folded into the lattice writer when delay_penalty > 0).  Returns px, py and what the backward needs."""
    B, T, C = am.shape
    S = lm.shape[1] - 1
    T1 = T if modified else T + 1
    cs = 1.0 - lm_only_scale - am_only_scale                    # :1342
    ls = lm_only_scale if lm_only_scale != 0.0 else 1.0e-20       # :1346-1349
    a_s = am_only_scale if am_only_scale != 0.0 else 1.0e-20
    amc = am.detach().contiguous(); lmc = lm.detach().contiguous()
    dev = amc.device
    am_probs = torch.empty_like(amc); lm_probs = torch.empty_like(lmc)
    am_max = torch.empty((B, T), dtype=torch.float32, device=dev)
    lm_max = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
    lm_sum = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
    px = torch.empty((B, S, T1), dtype=torch.float32, device=dev)
    py = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _stream_ptr(amc)
        _lib.call("ftr_rowmax_exp_sum_f32", _ptr(lmc), _ptr(lm_probs), _ptr(lm_max), _ptr(lm_sum),
                  B * (S + 1), C, st)                                                                   # :1276-1278
        inv = (1.0 / lm_sum).contiguous()                                                               # [B,S+1]
        ratio_sum = _colsum_weighted(lm_probs, inv, B * (S + 1), C, st)                                 # [C]
        count = float(B * (S + 1))
        if process_group is not None:
            # the mean runs over the rows of EVERY shard (rnnt_loss.py:1279-1280 on the global batch), and shards may hold
            # different numbers of rows (uneven batch split, every rank padded to its own S): the local row count travels
            # in the same all-reduce as the [C] sums and the reduced count -- a device scalar, no host read -- divides them
            packed = torch.cat((ratio_sum, torch.full((1,), count, dtype=torch.float32, device=dev)))
            torch.distributed.all_reduce(packed, group=process_group)
            ratio_sum, count = packed[:C], packed[C]
        u = (ratio_sum / count + _TINY).contiguous()                                                    # :1279-1280
        # am_probs, am_max and am_probs . u in one pass over am (:1265-1268, :1281-1286)
        am_dot = torch.empty((B * T,), dtype=torch.float32, device=dev)
        _lib.call("ftr_rowmax_exp_dot_f32", _ptr(amc), _ptr(am_probs), _ptr(am_max), _ptr(u), _ptr(am_dot), B * T, C, st)
        fused = _use_fused_builder(C)
        prod = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev) if fused else \
            _gemm(0, lm_probs, am_probs, B, T, S, C, st)                                                # :1270-1272
        amonly = (am_dot.log().reshape(B, T) + am_max).contiguous()                                     # :1281-1286
        ulog = u.log().contiguous()                                                                     # :1287
        lmonly = (lm_sum.log() + lm_max).contiguous()                                                   # :1288-1290
        if fused:
            _lib.call("ftr_smoothed_logprobs_fused_fwd_f32", _ptr(amc), _ptr(lmc), _ptr(symbols), _ptr(am_probs),
                      _ptr(lm_probs), _ptr(am_max), _ptr(lm_max), _ptr(lmonly), _ptr(amonly), _ptr(ulog), _ptr(boundary),
                      int(termination_symbol), float(delay_penalty), cs, ls, a_s, _ptr(px), _ptr(py), _ptr(prod),
                      B, T, S, C, int(modified), st)
        else:
            _lib.call("ftr_smoothed_logprobs_fwd_pen_f32", _ptr(amc), _ptr(lmc), _ptr(symbols), _ptr(prod), _ptr(am_max),
                      _ptr(lm_max), _ptr(lmonly), _ptr(amonly), _ptr(ulog), _ptr(boundary), int(termination_symbol),
                      float(delay_penalty), cs, ls, a_s, _ptr(px), _ptr(py), B, T, S, C, int(modified), st)
    saved = (am_probs, lm_probs, prod, symbols, boundary if boundary is not None else torch.empty(0), inv, u, am_dot)
    meta = (int(termination_symbol), int(modified), cs, ls, a_s, count, process_group)
    return px, py, saved, meta


def _smoothed_backward(saved, has_boundary, meta, gpx, gpy, scale=None, stride=0, mul=1.0):
    """Hand-written backward of the smoothed builder; gpx / gpy are d/d px, d/d py, multiplied on the fly by
    (scale ? scale[b * stride] : 1) * mul (the upstream gradient of the loss that owns the occupancies)."""
    am_probs, lm_probs, prod, symbols, boundary, inv, u, am_dot = saved
    if not has_boundary:
        boundary = None
    blank, modified, cs, ls, a_s, count, group = meta
    B, T, C = am_probs.shape
    S = lm_probs.shape[1] - 1
    dev = am_probs.device
    gpx = gpx.contiguous(); gpy = gpy.contiguous()
    W = torch.empty_like(prod)
    rsx = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
    rsy = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
    R = torch.empty((B, T), dtype=torch.float32, device=dev)
    d_am = torch.empty_like(am_probs); d_lm = torch.empty_like(lm_probs)
    with torch.cuda.device(dev):
        st = _stream_ptr(am_probs)
        _lib.call("ftr_smoothed_logprobs_bwd_w_scaled_f32", _ptr(gpx), _ptr(gpy), _ptr(scale), stride, mul, _ptr(prod),
                  _ptr(boundary), cs, _ptr(W), _ptr(rsx), _ptr(rsy), B, T, S, modified, st)
        dlmp = _gemm(1, W, am_probs, B, T, S, C, st)         # [B,S+1,C]
        if _use_fused_builder_bwd(T, C):                       # W^T lm_probs inside the d am kernel (opt-in)
            _lib.call("ftr_smoothed_logprobs_fused_bwd_am_f32", _ptr(gpx), _ptr(gpy), _ptr(scale), stride, mul, _ptr(prod),
                      _ptr(lm_probs), _ptr(am_probs), _ptr(symbols), _ptr(boundary), blank, cs, cs + a_s, _ptr(u),
                      _ptr(am_dot), a_s, _ptr(R), _ptr(d_am), B, T, S, C, modified, st)
        else:
            damp = _gemm(2, W, lm_probs, B, T, S, C, st)    # [B,T,C]
            _lib.call("ftr_smoothed_logprobs_bwd_am_scaled_f32", _ptr(gpx), _ptr(gpy), _ptr(scale), stride, mul, _ptr(damp),
                      _ptr(am_probs), _ptr(symbols), _ptr(boundary), blank, cs + a_s, _ptr(u), _ptr(am_dot), a_s, _ptr(R),
                      _ptr(d_am), B, T, S, C, modified, st)
        # d u: through amonly_norm and through ulog
        du = _colsum_weighted(am_probs, R, B * T, C, st)
        gul = torch.zeros((C,), dtype=torch.float32, device=dev)
        if S > 0:
            gul.index_add_(0, _i64(symbols).reshape(-1), rsx[:, :S].reshape(-1))
        gul[blank] += rsy.sum()
        du = du + a_s * gul / u
        if group is not None:
            torch.distributed.all_reduce(du, group=group)
        gu = (du / count).contiguous()
        dotq = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        _lib.call("ftr_rowdot_f32", _ptr(lm_probs), _ptr(gu), _ptr(dotq), B * (S + 1), C, st)
        dotq = dotq * inv
        arow = ((-ls) * (rsx + rsy) - dotq) * inv
        arow = arow.contiguous()
        _lib.call("ftr_smoothed_logprobs_bwd_lm_f32", _ptr(dlmp), _ptr(lm_probs), _ptr(symbols), _ptr(rsx),
                  _ptr(rsy), blank, cs + ls, _ptr(arow), _ptr(inv), _ptr(gu), _ptr(d_lm), B, S, C, st)
    return d_lm, d_am


class _SmoothedLogprobs(torch.autograd.Function):
    """get_rnnt_logprobs_smoothed (+ fix_for_boundary) for regular/modified on the native builder kernels.

    Forward (rnnt_loss.py:1265-1365): out = cs (x - normalizers) + ls (lm - lmonly_norm) + as (am + ulog - amonly_norm)
    with lmonly_norm = log(rowsum lm_probs) + lm_max, u = mean_{b,s}(lm_probs / rowsum) + tiny, ulog = log u,
    amonly_norm = log(am_probs . u) + am_max.  The lattice-sized work is in ftr_smoothed_logprobs_*; the batch
    statistics ([C], [B,S+1] and [B,T] vectors, two matvecs) are torch ops on the same stream.

    Backward, with gx = gpx masked where the forward wrote -inf, gy = gpy, rsx/rsy their sums over t and
    colsum over s:
      normalizers : W = -cs (gx+gy)/(prod+tiny), two GEMMs (as in _SimpleLogprobs)
      direct      : am column sym/blank gets (cs+as) g, lm column sym/blank gets (cs+ls) g
      lmonly_norm : d lm += -ls (rsx+rsy) * lm_probs / rowsum
      amonly_norm : R[b,t] = -as colsum(gx+gy) / (am_probs . u);  d am += R am_probs u;  d u += R^T am_probs
      ulog        : d u += as (sum of rsx by symbol + sum rsy at blank) / u
      u           : gu = d u / N;  d lm += lm_probs/rowsum * (gu - (lm_probs/rowsum) . gu)
    """

    @staticmethod
    def forward(ctx, lm, am, symbols, termination_symbol, boundary, modified, lm_only_scale, am_only_scale,
                process_group):
        px, py, saved, meta = _smoothed_forward(lm, am, symbols, termination_symbol, boundary, modified, lm_only_scale,
                                                am_only_scale, process_group, 0.0)
        ctx.save_for_backward(*saved)
        ctx.has_boundary = boundary is not None
        ctx.meta = meta
        return px, py

    @staticmethod
    def backward(ctx, gpx, gpy):
        d_lm, d_am = _smoothed_backward(ctx.saved_tensors, ctx.has_boundary, ctx.meta, gpx, gpy)
        return d_lm, d_am, None, None, None, None, None, None, None


class _SmoothedLoss(torch.autograd.Function):
    """rnnt_loss_smoothed for regular/modified as ONE graph node, like _SimpleLoss: smoothed builder (penalty folded
    in) + GEMM, recursion forward + backward (occupancies), native loss reduction; backward() feeds the occupancies to
    the builder's backward kernels with the upstream gradient folded in on the fly."""

    @staticmethod
    def forward(ctx, lm, am, symbols, termination_symbol, boundary, modified, lm_only_scale, am_only_scale,
                process_group, delay_penalty, code, want_occupancies):
        px, py, saved, meta = _smoothed_forward(lm, am, symbols, termination_symbol, boundary, modified, lm_only_scale,
                                                am_only_scale, process_group, delay_penalty)
        need = bool(want_occupancies) or ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if need:      # the loss tail rides along with the recursion's backward launch
            ans, px_grad, py_grad, loss = mi_forward_backward(px, py, boundary, True, ans_grad_is_one=True, loss_code=code)
        else:
            ans, px_grad, py_grad = mi_forward_backward(px, py, boundary, False, ans_grad_is_one=True)
            loss = _negated_reduce_native(ans, code)
        if need:
            ctx.save_for_backward(*saved, px_grad, py_grad)
        else:
            px_grad = torch.zeros_like(px); py_grad = torch.zeros_like(py)
        del px, py
        ctx.has_boundary = boundary is not None
        ctx.meta = meta
        ctx.code = int(code)
        ctx.mark_non_differentiable(px_grad, py_grad)
        ctx.set_materialize_grads(False)          # no zero tensors for the two occupancy outputs in backward
        return loss, px_grad, py_grad

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        if g_loss is None:
            return (None,) * 12
        *saved, px_grad, py_grad = ctx.saved_tensors
        B = saved[0].shape[0]
        scale, stride, mul = _upstream_scale(g_loss, ctx.code, B)
        d_lm, d_am = _smoothed_backward(tuple(saved), ctx.has_boundary, ctx.meta, px_grad, py_grad, scale, stride, mul)
        return d_lm, d_am, None, None, None, None, None, None, None, None, None, None


def get_rnnt_logprobs_smoothed(
    lm: torch.Tensor,
    am: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    lm_only_scale: float = 0.1,
    am_only_scale: float = 0.1,
    boundary: Optional[torch.Tensor] = None,
    rnnt_type: str = "regular",
    process_group=None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """rnnt_loss.py:1132-1367, native (HIP) builder, differentiable w.r.t. lm and am.  ``process_group``
    (extension, default None = local batch only): when the batch is sharded over ranks, the batch-wide
    ``unigram_lm`` mean (rnnt_loss.py:1279-1280) is all-reduced (one [C] vector over RCCL forward, one backward)
    so every shard sees the global-batch value."""
    _check_type(rnnt_type)
    _require_gpu(am, "am"); _require_gpu(lm, "lm")
    if am.dtype != torch.float32 or lm.dtype != torch.float32:
        raise TypeError("am and lm must be float32")
    B, T, C = am.shape
    S = lm.shape[1] - 1
    if lm.shape[0] != B or lm.shape[2] != C:
        raise ValueError(f"lm {tuple(lm.shape)} and am {tuple(am.shape)} disagree")
    symbols = torch.as_tensor(symbols, device=am.device)
    if tuple(symbols.shape) != (B, S):
        raise ValueError(f"symbols must have shape {(B, S)}, got {tuple(symbols.shape)}")
    if not 0 <= int(termination_symbol) < C:
        raise ValueError(f"termination_symbol {termination_symbol} not in [0, {C})")
    symbols = symbols.to(torch.int32).contiguous()
    boundary = _as_boundary(boundary, B, am.device)
    px, py = _SmoothedLogprobs.apply(lm, am, symbols, termination_symbol, boundary, rnnt_type != "regular",
                                     float(lm_only_scale), float(am_only_scale), process_group)
    if rnnt_type == "constrained":
        px = px + py[:, 1:, :]                                                                              # :1362-1363
    return px, py


def rnnt_loss_smoothed(
    lm: torch.Tensor,
    am: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    lm_only_scale: float = 0.1,
    am_only_scale: float = 0.1,
    boundary: Optional[torch.Tensor] = None,
    rnnt_type: str = "regular",
    delay_penalty: float = 0.0,
    reduction: Optional[str] = "mean",
    calc_gradients: bool = False,
    process_group=None,
) -> Union[Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], torch.Tensor]:
    """rnnt_loss.py:1369-1494.  regular / modified: one fused node (no framework-side pass over a lattice)."""
    _check_type(rnnt_type)
    boundary = _as_boundary(boundary, am.shape[0], am.device)
    if rnnt_type == "constrained":   # the penalty applies after px += py[:, 1:, :]  (:1362-1363, :1461-1478)
        px, py = get_rnnt_logprobs_smoothed(lm=lm, am=am, symbols=symbols, termination_symbol=termination_symbol,
                                            lm_only_scale=lm_only_scale, am_only_scale=am_only_scale,
                                            boundary=boundary, rnnt_type=rnnt_type, process_group=process_group)
        px = _apply_delay_penalty(px, boundary, rnnt_type, delay_penalty)
        return _drive(px, py, boundary, reduction, calc_gradients)
    code = _reduction_code(reduction)
    symbols_i = _check_simple_inputs(lm, am, symbols, termination_symbol)
    pen = float(delay_penalty) if delay_penalty > 0.0 else 0.0
    loss, px_grad, py_grad = _SmoothedLoss.apply(lm, am, symbols_i, termination_symbol, boundary, rnnt_type != "regular",
                                                 float(lm_only_scale), float(am_only_scale), process_group, pen, code,
                                                 bool(calc_gradients))
    return (loss, (px_grad, py_grad)) if calc_gradients else loss
