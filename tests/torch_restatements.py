"""Op-by-op torch restatements of the reference's px/py builders (the shape the reference has them in:
tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:163-223, 340-452, 814-851, 1132-1367).  TEST INFRASTRUCTURE ONLY: they run
on CPU tensors in float64 so that torch autograd can check the hand-written backward kernels; nothing in the
product package imports this file."""
from typing import Optional, Tuple

import torch

from tf_fast_rnnt.rnnt_loss import _check_type, _i64, _NEG_INF, _TINY, fix_for_boundary


def _normalizers(lm: torch.Tensor, am: torch.Tensor):
    """rnnt_loss.py:175-186."""
    am_max = am.max(dim=2, keepdim=True).values            # [B,T,1]
    lm_max = lm.max(dim=2, keepdim=True).values            # [B,S+1,1]
    am_probs = (am - am_max).exp()
    lm_probs = (lm - lm_max).exp()
    normalizers = (torch.matmul(lm_probs, am_probs.transpose(1, 2)) + _TINY).log()
    normalizers = normalizers + lm_max + am_max.transpose(1, 2)   # [B,S+1,T]
    return normalizers, am_max, lm_max, am_probs, lm_probs


def get_rnnt_logprobs_torch(lm, am, symbols, termination_symbol, rnnt_type="regular", boundary=None):
    """The same function op by op in torch (the shape the reference has it in, rnnt_loss.py:163-223).  Independent restatement for the tests."""
    _check_type(rnnt_type)
    B, T, C = am.shape
    S = lm.shape[1] - 1
    if tuple(symbols.shape) != (B, S):
        raise ValueError(f"symbols must have shape {(B, S)}, got {tuple(symbols.shape)}")
    sym = _i64(symbols)
    normalizers, *_ = _normalizers(lm, am)
    # px_am[b,s,t] = am[b,t,symbols[b,s]]                                   (:187-192)
    px_am = torch.gather(am.transpose(1, 2), 1, sym.unsqueeze(2).expand(B, S, T))
    if rnnt_type == "regular":
        px_am = torch.cat((px_am, torch.full((B, S, 1), _NEG_INF, dtype=am.dtype, device=am.device)), dim=2)
    px_lm = torch.gather(lm[:, :S, :], 2, sym.unsqueeze(2))                  # [B,S,1]   (:204-207)
    px = px_am + px_lm
    if rnnt_type == "regular":
        px = px - torch.cat((normalizers, torch.zeros((B, S + 1, 1), dtype=am.dtype, device=am.device)), dim=2)[:, :S, :]
    else:
        px = px - normalizers[:, :S, :]
    py_am = am[:, :, termination_symbol].unsqueeze(1)                        # [B,1,T]
    py_lm = lm[:, :, termination_symbol].unsqueeze(2)                        # [B,S+1,1]
    py = py_am + py_lm - normalizers
    if rnnt_type == "regular":
        px = fix_for_boundary(px, boundary)
    elif rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return px, py


def get_rnnt_logprobs_joint_torch(
    logits: torch.Tensor,
    symbols: torch.Tensor,
    termination_symbol: int,
    boundary: Optional[torch.Tensor] = None,
    rnnt_type: str = "regular",
) -> Tuple[torch.Tensor, torch.Tensor]:
    """rnnt_loss.py:340-452 op by op in torch.  Not on the product path: independent restatement for the tests."""
    _check_type(rnnt_type)
    B, T, S1, C = logits.shape
    S = S1 - 1
    sym = _i64(symbols)
    normalizers = torch.logsumexp(logits, dim=3).permute(0, 2, 1)            # [B,S+1,T]
    px = torch.gather(logits[:, :, :S, :], 3, sym.reshape(B, 1, S, 1).expand(B, T, S, 1)).squeeze(-1)
    px = px.permute(0, 2, 1)                                                  # [B,S,T]
    if rnnt_type == "regular":
        px = torch.cat((px, torch.full((B, S, 1), _NEG_INF, dtype=logits.dtype, device=logits.device)), dim=2)
        px = px - torch.cat((normalizers, torch.zeros((B, S + 1, 1), dtype=logits.dtype, device=logits.device)), dim=2)[:, :S, :]
    else:
        px = px - normalizers[:, :S, :]
    py = logits[:, :, :, termination_symbol].permute(0, 2, 1) - normalizers
    if rnnt_type == "regular":
        px = fix_for_boundary(px, boundary)
    elif rnnt_type == "constrained":
        px = px + py[:, 1:, :]
    return px.contiguous(), py.contiguous()


def roll_by_shifts(src: torch.Tensor, shifts: torch.Tensor) -> torch.Tensor:
    """rnnt_loss.py:814-851: out[b,t,i] = src[b,t,(i - shifts[b,t]) % S]."""
    B, T, S = src.shape
    index = (torch.arange(S, device=src.device).reshape(1, 1, S) - _i64(shifts).reshape(B, T, 1)) % S
    return torch.gather(src, 2, index)


def get_rnnt_logprobs_smoothed_torch(
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
    """rnnt_loss.py:1132-1367 op by op in torch (the shape the reference has it in).  Not on the product path:
    an independent restatement for the tests (autograd in float64 checks the hand-written backward)."""
    _check_type(rnnt_type)
    B, T, C = am.shape
    S = lm.shape[1] - 1
    sym = _i64(symbols)
    normalizers, am_max, lm_max, am_probs, lm_probs = _normalizers(lm, am)
    lmonly_normalizers = lm_probs.sum(dim=2, keepdim=True)                    # [B,S+1,1]   (:1276-1278)
    ratio_sum = (lm_probs / lmonly_normalizers).sum(dim=(0, 1), keepdim=True)  # [1,1,C]
    count = float(B * (S + 1))
    if process_group is not None:
        from tf_fast_rnnt.distributed import all_reduce_sum_differentiable
        ratio_sum = all_reduce_sum_differentiable(ratio_sum, process_group)
        cnt = torch.tensor([count], dtype=torch.float64)      # rows of every shard (shards may be uneven)
        torch.distributed.all_reduce(cnt, group=process_group)
        count = float(cnt.item())
    unigram_lm = ratio_sum / count + _TINY                                    # (:1279-1280)
    amonly_normalizers = (torch.mv(am_probs.reshape(-1, C), unigram_lm.reshape(C)).log().reshape(B, T, 1) + am_max)
    amonly_normalizers = amonly_normalizers.transpose(1, 2)                   # [B,1,T]     (:1281-1286)
    unigram_lm = unigram_lm.log()
    lmonly_normalizers = lmonly_normalizers.log() + lm_max                    # [B,S+1,1]   (:1288-1290)

    px_am = torch.gather(am.transpose(1, 2), 1, sym.unsqueeze(2).expand(B, S, T))
    regular = rnnt_type == "regular"
    if regular:
        px_am = torch.cat((px_am, torch.full((B, S, 1), _NEG_INF, dtype=am.dtype, device=am.device)), dim=2)
    px_lm = torch.gather(lm[:, :S, :], 2, sym.unsqueeze(2))                   # [B,S,1]
    px_lm_unigram = unigram_lm.reshape(-1)[sym].unsqueeze(2)                   # [B,S,1]     (:1319-1321)
    px = px_am + px_lm
    if regular:
        px = px - torch.cat((normalizers, torch.zeros((B, S + 1, 1), dtype=am.dtype, device=am.device)), dim=2)[:, :S, :]
        px_amonly = (px_am + px_lm_unigram) - torch.cat(
            (amonly_normalizers, torch.zeros((B, 1, 1), dtype=am.dtype, device=am.device)), dim=2)
    else:
        px = px - normalizers[:, :S, :]
        px_amonly = (px_am + px_lm_unigram) - amonly_normalizers
    px_lmonly = px_lm - lmonly_normalizers[:, :S, :]

    py_am = am[:, :, termination_symbol].unsqueeze(1)
    py_lm = lm[:, :, termination_symbol].unsqueeze(2)
    py = py_am + py_lm - normalizers
    py_lm_unigram = unigram_lm[0][0][termination_symbol]
    py_amonly = py_am + py_lm_unigram - amonly_normalizers
    py_lmonly = py_lm - lmonly_normalizers

    combined_scale = 1.0 - lm_only_scale - am_only_scale
    if lm_only_scale == 0.0:
        lm_only_scale = 1.0e-20
    if am_only_scale == 0.0:
        am_only_scale = 1.0e-20
    px_interp = px * combined_scale + px_lmonly * lm_only_scale + px_amonly * am_only_scale
    py_interp = py * combined_scale + py_lmonly * lm_only_scale + py_amonly * am_only_scale
    if regular:
        px_interp = fix_for_boundary(px_interp, boundary)
    elif rnnt_type == "constrained":
        px_interp = px_interp + py_interp[:, 1:, :]
    return px_interp, py_interp


