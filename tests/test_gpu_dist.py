"""N > 1 on the GPU box: two ranks (both on cuda:0, gloo rendezvous on 127.0.0.1) run the sharded smoothed + pruned
path through the native kernels; the sharded result must equal the single-process full-batch result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tf-fast-rnnt_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tf_fast_rnnt as ft
    from tf_fast_rnnt.distributed import reduce_loss, shard_batch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(77)
    B, T, S, C, r = 7, 40, 11, 16, 4          # 7 utterances over 2 ranks: uneven shards (4 + 3)
    am_f = torch.randn((B, T, C), generator=g).to(dev); lm_f = torch.randn((B, S + 1, C), generator=g).to(dev)
    sym_f = torch.randint(0, C - 1, (B, S), generator=g, dtype=torch.int32).to(dev)
    bd_f = torch.zeros((B, 4), dtype=torch.int32); bd_f[:, 2] = S; bd_f[:, 3] = T
    bd_f[1, 2] = 7; bd_f[1, 3] = 29; bd_f[4, 3] = 33
    bd_f = bd_f.to(dev)

    def pipeline(am, lm, sym, bd, group):
        am = am.clone().requires_grad_(True); lm = lm.clone().requires_grad_(True)
        loss, (gx, gy) = ft.rnnt_loss_smoothed(lm, am, sym, C - 1, 0.1, 0.2, bd, "regular", 0.0, "none", True,
                                               process_group=group)
        ranges = ft.get_rnnt_prune_ranges(gx, gy, bd, r)
        am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
        ploss = ft.rnnt_loss_pruned(torch.tanh(am_p + lm_p), sym, ranges, C - 1, bd, "regular", 0.0, "none")
        return am, lm, loss, ploss

    lo, hi = shard_batch(B, rank, world)
    am, lm, loss, ploss = pipeline(am_f[lo:hi], lm_f[lo:hi], sym_f[lo:hi], bd_f[lo:hi], dist.group.WORLD)
    total = reduce_loss(0.5 * loss + ploss, "sum")          # global value, local gradient
    total.backward()
    # single-process full batch on the same device
    am2, lm2, loss2, ploss2 = pipeline(am_f, lm_f, sym_f, bd_f, None)
    total2 = (0.5 * loss2 + ploss2).sum()
    total2.backward()
    # the unigram couples the shards: d lm of a local utterance also collects the other shard's dependence on it,
    # which the all-reduced d unigram provides
    ok = dict(
        loss=torch.allclose(loss, loss2[lo:hi], rtol=2e-5, atol=1e-4),
        ploss=torch.allclose(ploss, ploss2[lo:hi], rtol=2e-5, atol=1e-4),
        total=abs(total.item() - total2.item()) <= 2e-5 * abs(total2.item()),
        am=torch.allclose(am.grad, am2.grad[lo:hi], rtol=1e-3, atol=2e-5),
        lm=torch.allclose(lm.grad, lm2.grad[lo:hi], rtol=1e-3, atol=2e-5),
    )
    q.put((rank, ok, float((lm.grad - lm2.grad[lo:hi]).abs().max()), float(lm2.grad.abs().max())))
    dist.destroy_process_group()


def test_sharded_smoothed_pruned_two_ranks(ft, dev):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, q)) for rk in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err, scale in res:
        assert all(ok.values()), (rank, ok, err, scale)
