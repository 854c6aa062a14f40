"""Batch-sharded multi-GPU use of the loss (the reference has no distributed code at all).

Every kernel of the path is independent per utterance (``b`` only indexes,
mutual_information_cuda.cu:247-248), so the batch is split into contiguous slices, one per rank / GPU
(one process per GPU, ``torch.distributed`` with backend "nccl" = RCCL over xGMI).  The data path needs
no collective; the only exchanges are

* the batch reduction of the loss (rnnt_loss.py:333,544-546,1124-1126,1487-1489): one all-reduce of a
  scalar (plus the utterance count for "mean");
* for ``rnnt_loss_smoothed`` the batch-wide ``unigram_lm`` mean (rnnt_loss.py:1279-1280): one [C]
  vector forward and its gradient backward (``all_reduce_sum_differentiable``).

Both messages are <= 4 KB: latency-bound on xGMI, so they are issued once per step, never bucketed.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def all_reduce_sum_differentiable(x: torch.Tensor, group=None) -> torch.Tensor:
    """sum over ranks, with the transpose (another sum over ranks) in backward."""
    return _AllReduceSum.apply(x, group)


def shard_batch(B: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def reduce_loss(local_loss_none: torch.Tensor, reduction: str = "mean", group=None,
                grad_averaging: bool = False) -> torch.Tensor:
    """Combine per-utterance losses (reduction="none" output of any loss driver, this rank's shard) into
    the value the single-device call would return for the whole batch: exactly one all-reduce of 2 floats.

    The returned scalar has the global VALUE on every rank and the LOCAL gradient: d/d(loss of a local
    utterance) is 1 ("sum") or 1/global_count ("mean").

    WHAT THE CALLER'S GRADIENT EXCHANGE MUST DO.  With ``grad_averaging=False`` (default) the parameter gradients
    must be SUMMED over ranks (a plain all-reduce SUM): the sum of the local gradients is the single-device
    gradient (the replicated scalar is not counted world_size times).  ``torch.nn.parallel.DistributedDataParallel``
    AVERAGES gradients instead; pass ``grad_averaging=True`` there: the local gradient is multiplied by the world
    size (the value is unchanged) so that the average over ranks is again the single-device gradient.  The same
    factor reaches the smoothed builder's all-reduced unigram gradient (``get_rnnt_logprobs_smoothed(process_group=)``),
    which is linear in the upstream gradient."""
    if reduction not in ("mean", "sum"):
        raise ValueError("reduce_loss supports 'mean' and 'sum'")
    local_sum = local_loss_none.sum()
    packed = torch.stack((local_sum.detach(), local_sum.new_tensor(float(local_loss_none.numel()))))
    world = 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    grad_part = local_sum * float(world) if (grad_averaging and world > 1) else local_sum
    total = grad_part + (packed[0] - grad_part.detach())      # global value, local (optionally world-scaled) gradient
    return total if reduction == "sum" else total / packed[1]
