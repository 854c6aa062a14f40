"""GPU parity of the pruning pipeline and of the loss drivers against the oracle, through the package
(ctypes -> C ABI -> HIP).  Integer outputs are compared bit-exactly; floats as in test_gpu_mi.py."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_parity, max_rel, reference_test_recipe, synthetic

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _occupancies(oracle, d, **kw):
    loss, (gx, gy) = oracle.rnnt_loss_smoothed(d["lm"], d["am"], d["symbols"], d["termination_symbol"],
                                               lm_only_scale=0.1, am_only_scale=0.2, boundary=d["boundary"],
                                               reduction="none", delay_penalty=0.2, calc_gradients=True, **kw)
    return loss, gx, gy


@pytest.mark.parametrize("cfg", [(1234, 2, 10, 7, 4), (12345, 2, 200, 50, 50), (7, 5, 90, 33, 20)])
def test_prune_ranges_bit_exact(ft, dev, oracle, cfg):
    """simple_rnnt_loss_test.py:291-336 scenario: occupancies from the smoothed loss, r = 2..S+3."""
    d = reference_test_recipe(*cfg)
    _, gx, gy = _occupancies(oracle, d)
    S = d["S"]
    for r in list(range(2, min(S + 4, 12))) + [S, S + 1, S + 3]:
        got = ft.get_rnnt_prune_ranges(_t(gx, dev), _t(gy, dev), _t(d["boundary"], dev), r).cpu().numpy()
        want = oracle.get_rnnt_prune_ranges(gx, gy, d["boundary"], r)
        assert got.dtype == np.int32 and got.shape == want.shape
        assert np.array_equal(got, want), f"r={r}"
        # documented properties (rnnt_loss.py:673-677)
        s0 = got[:, :, 0]
        assert (np.diff(s0, axis=1) >= 0).all() and (s0 >= 0).all()
        assert (np.diff(s0, axis=1) <= max(got.shape[2] - 1, 0)).all()


def test_prune_ranges_modified_and_ties(ft, dev, oracle):
    rng = np.random.default_rng(3)
    B, S, T = 3, 9, 17
    gy = rng.random((B, S + 1, T)).astype(np.float32)
    gx = rng.random((B, S, T)).astype(np.float32)          # modified: T1 == T
    gy[0] = 0.25                                            # exact ties everywhere -> first maximum
    gx[0] = 0.25
    bd = np.array([[0, 0, S, T], [0, 0, 4, 9], [0, 0, 2, 17]], dtype=np.int32)
    for r in (1, 2, 3, 5, 20):
        got = ft.get_rnnt_prune_ranges(_t(gx, dev), _t(gy, dev), _t(bd, dev), r).cpu().numpy()
        assert np.array_equal(got, oracle.get_rnnt_prune_ranges(gx, gy, bd, r)), r
        assert np.array_equal(got, oracle.get_rnnt_prune_ranges_numpy(gx, gy, bd, r)), r


def test_adjust_lower_bound_opwise(ft, dev, oracle):
    """_adjust_pruning_lower_bound built from the native cummin op, like the reference (rnnt_loss.py:587-641)."""
    from tf_fast_rnnt.rnnt_loss import _adjust_pruning_lower_bound, _monotonic_lower_bound
    x = np.array([[12, 18, 5, 4, 18, 17], [11, 14, 14, 3, 10, 4], [19, 3, 8, 13, 7, 19]], dtype=np.int32)
    want = np.array([[4, 4, 4, 4, 17, 17], [3, 3, 3, 3, 4, 4], [3, 3, 7, 7, 7, 19]], dtype=np.int32)  # rnnt_loss.py:568-574
    assert np.array_equal(_monotonic_lower_bound(_t(x, dev)).cpu().numpy(), want)
    v = np.array([0, 2, 1, 3, 6, 5, 8], dtype=np.int32)                                                  # rnnt_loss.py:561-563
    assert np.array_equal(_monotonic_lower_bound(_t(v, dev)).cpu().numpy(), [0, 1, 1, 3, 5, 5, 8])
    rng = np.random.default_rng(0)
    s = rng.integers(0, 30, (4, 50)).astype(np.int32)
    for r in (2, 3, 7):
        assert np.array_equal(_adjust_pruning_lower_bound(_t(s, dev), r).cpu().numpy(), oracle.adjust_pruning_lower_bound(s, r))


@pytest.mark.parametrize("C", [16, 50, 7])
def test_do_pruning_bit_exact_and_grad(ft, dev, oracle, C):
    d = synthetic(1, 3, 21, 9, C, ragged=True)
    _, gx, gy = _occupancies(oracle, d)
    ranges = oracle.get_rnnt_prune_ranges(gx, gy, d["boundary"], 4)
    am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
    am_p, lm_p = ft.do_rnnt_pruning(am, lm, _t(ranges, dev))
    o_am, o_lm = oracle.do_rnnt_pruning(d["am"], d["lm"], ranges)
    assert np.array_equal(am_p.detach().cpu().numpy(), o_am) and np.array_equal(lm_p.detach().cpu().numpy(), o_lm)
    w1 = torch.randn_like(am_p); w2 = torch.randn_like(lm_p)
    ((am_p * w1).sum() + (lm_p * w2).sum()).backward()
    np.testing.assert_allclose(am.grad.cpu().numpy(), w1.sum(dim=2).cpu().numpy(), rtol=1e-6)
    want = np.zeros_like(d["lm"])
    w2n = w2.cpu().numpy()
    B, T, r = ranges.shape
    for b in range(B):
        for t in range(T):
            for k in range(r):
                want[b, ranges[b, t, k]] += w2n[b, t, k]
    np.testing.assert_allclose(lm.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_do_pruning_am_is_a_broadcast_view_and_the_c_abi_still_materialises(ft, dev, oracle):
    """am_pruned[b,t,k,:] = am[b,t,:] (rnnt_loss.py:803, a tf.broadcast_to).  The package returns it as a stride-0 view over
    s_range (no B*T*r*C write), with the reference's values, shape and gradient; .contiguous() gives the dense tensor.  The
    C ABI writes both dense outputs when given two pointers and only the gather when am_pruned is NULL."""
    from tf_fast_rnnt import _lib
    from tf_fast_rnnt.mutual_information import _ptr
    d = synthetic(2, 3, 30, 11, 24, ragged=True)
    _, gx, gy = _occupancies(oracle, d)
    ranges = oracle.get_rnnt_prune_ranges(gx, gy, d["boundary"], 4)
    o_am, o_lm = oracle.do_rnnt_pruning(d["am"], d["lm"], ranges)
    am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
    rg = _t(ranges, dev)
    am_p, lm_p = ft.do_rnnt_pruning(am, lm, rg)
    B, T, r = ranges.shape
    C = d["am"].shape[2]
    assert tuple(am_p.shape) == (B, T, r, C) and am_p.stride(2) == 0 and lm_p.is_contiguous()
    assert np.array_equal(am_p.detach().contiguous().cpu().numpy(), o_am) and np.array_equal(lm_p.detach().cpu().numpy(), o_lm)
    # a joiner consumes it by broadcasting; the gradient is the sum over s_range of what comes back
    w = torch.randn((B, T, r, C), device=dev)
    (torch.tanh(am_p + lm_p) * w).sum().backward()
    g = (w * (1 - torch.tanh(_t(o_am, dev) + _t(o_lm, dev)) ** 2))
    np.testing.assert_allclose(am.grad.cpu().numpy(), g.sum(dim=2).cpu().numpy(), rtol=1e-5, atol=1e-6)
    # the C ABI: dense am_pruned when asked for, untouched when NULL
    dense_am = torch.full((B, T, r, C), -7.0, device=dev); dense_lm = torch.empty((B, T, r, C), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    S1 = d["lm"].shape[1]
    _lib.call("ftr_do_pruning_f32", _ptr(am.detach()), _ptr(lm.detach()), _ptr(rg), _ptr(dense_am), _ptr(dense_lm), B, T, S1, C, r, st)
    assert np.array_equal(dense_am.cpu().numpy(), o_am) and np.array_equal(dense_lm.cpu().numpy(), o_lm)
    only_lm = torch.empty((B, T, r, C), device=dev)
    _lib.call("ftr_do_pruning_f32", _ptr(am.detach()), _ptr(lm.detach()), _ptr(rg), None, _ptr(only_lm), B, T, S1, C, r, st)
    assert np.array_equal(only_lm.cpu().numpy(), o_lm)


@pytest.mark.parametrize("rnnt_type", ["regular", "modified", "constrained"])
def test_pruned_logprobs_exact_pattern(ft, dev, oracle, rnnt_type):
    d = reference_test_recipe(1234, 2, 40, 12, 16)
    _, gx, gy = _occupancies(oracle, d)
    r = 4
    ranges = oracle.get_rnnt_prune_ranges(gx, gy, d["boundary"], r)
    am_p, lm_p = oracle.do_rnnt_pruning(d["am"], d["lm"], ranges)
    logits = (1.0 / (1.0 + np.exp(-(am_p + lm_p)))).astype(np.float32)
    px, py = ft.get_rnnt_logprobs_pruned(_t(logits, dev), _t(d["symbols"], dev), _t(ranges, dev), d["termination_symbol"],
                                         _t(d["boundary"], dev), rnnt_type)
    o_px, o_py = oracle.get_rnnt_logprobs_pruned(logits, d["symbols"], ranges, d["termination_symbol"], d["boundary"], rnnt_type)
    px = px.cpu().numpy(); py = py.cpu().numpy()
    assert px.shape == o_px.shape and py.shape == o_py.shape
    assert np.array_equal(np.isneginf(px), np.isneginf(o_px)) and np.array_equal(np.isneginf(py), np.isneginf(o_py))
    fin = np.isfinite(o_px)
    np.testing.assert_allclose(px[fin], o_px[fin], rtol=1e-5, atol=1e-6)
    fin = np.isfinite(o_py)
    np.testing.assert_allclose(py[fin], o_py[fin], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cfg", [(1234, 2, 10, 7, 4), (12345, 2, 200, 50, 50)])
def test_reference_scenario(ft, dev, oracle, cfg):
    """The scenario of the reference's only active test (simple_rnnt_loss_test.py:256-369), with
    assertions instead of prints: unpruned rnnt_loss, smoothed loss with occupancies, then for a sweep of
    s_range: ranges -> gather -> sigmoid joiner -> pruned loss and its gradient w.r.t. logits."""
    d = reference_test_recipe(*cfg)
    am, lm, sym, bd = (_t(d[k], dev) for k in ("am", "lm", "symbols", "boundary"))
    blank = d["termination_symbol"]
    # unpruned rnnt_loss on joiner logits
    logits_np = 1.0 / (1.0 + np.exp(-(d["am"][:, :, None, :] + d["lm"][:, None, :, :])))
    logits = _t(logits_np.astype(np.float32), dev)
    loss = ft.rnnt_loss(logits=logits, symbols=sym, termination_symbol=blank, boundary=bd, reduction="mean", delay_penalty=0.2)
    o_loss = oracle.rnnt_loss(logits_np.astype(np.float32), d["symbols"], blank, d["boundary"], reduction="mean", delay_penalty=0.2)
    np.testing.assert_allclose(loss.item(), o_loss, rtol=1e-4)
    # smoothed
    sl, (pxg, pyg) = ft.rnnt_loss_smoothed(lm=lm, am=am, symbols=sym, termination_symbol=blank, boundary=bd,
                                           lm_only_scale=0.1, am_only_scale=0.2, reduction="none", delay_penalty=0.2,
                                           calc_gradients=True)
    o_sl, o_gx, o_gy = _occupancies(oracle, d)
    np.testing.assert_allclose(sl.cpu().numpy(), o_sl, rtol=1e-4)
    assert max_rel(pxg.cpu().numpy(), o_gx) <= 1e-4 and max_rel(pyg.cpu().numpy(), o_gy) <= 1e-4
    S = d["S"]
    for r in sorted(set([2, 3, 5, min(8, S), S, S + 1])):
        ranges = oracle.get_rnnt_prune_ranges(o_gx, o_gy, d["boundary"], r)
        got_ranges = ft.get_rnnt_prune_ranges(_t(o_gx, dev), _t(o_gy, dev), bd, r)
        assert np.array_equal(got_ranges.cpu().numpy(), ranges)
        am_p, lm_p = ft.do_rnnt_pruning(am=am, lm=lm, ranges=got_ranges)
        lg = torch.sigmoid(am_p + lm_p).detach().requires_grad_(True)
        for reduction in ("mean", "sum"):
            lg.grad = None
            pl = ft.rnnt_loss_pruned(logits=lg, symbols=sym, ranges=got_ranges, termination_symbol=blank, boundary=bd,
                                     reduction=reduction, delay_penalty=0.2, calc_gradients=True)
            pl.backward()
            o_pl, o_g = oracle.rnnt_loss_pruned_grad(lg.detach().cpu().numpy(), d["symbols"], ranges, blank, d["boundary"],
                                                     delay_penalty=0.2, reduction=reduction)
            np.testing.assert_allclose(pl.item(), o_pl, rtol=1e-4)
            _, o_g64 = oracle.rnnt_loss_pruned_grad(lg.detach().cpu().numpy(), d["symbols"], ranges, blank, d["boundary"],
                                                    delay_penalty=0.2, reduction=reduction, dtype=np.float64)
            assert_parity(lg.grad.cpu().numpy(), o_g, o_g64, what=f"pruned grad r={r} {reduction}")


@pytest.mark.parametrize("rnnt_type", ["regular", "modified", "constrained"])
def test_simple_loss_types_and_grads(ft, dev, oracle, rnnt_type):
    d = synthetic(5, 3, 24, 8, 12, ragged=True)
    am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
    loss, (pxg, pyg) = ft.rnnt_loss_simple(lm=lm, am=am, symbols=_t(d["symbols"], dev), termination_symbol=d["termination_symbol"],
                                           boundary=_t(d["boundary"], dev), rnnt_type=rnnt_type, reduction="none",
                                           delay_penalty=0.1, calc_gradients=True)
    o_loss, (o_gx, o_gy) = oracle.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], d["termination_symbol"], d["boundary"],
                                                   rnnt_type, 0.1, "none", True)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), o_loss, rtol=1e-4)
    assert pxg.shape == o_gx.shape
    assert max_rel(pxg.cpu().numpy(), o_gx) <= 1e-4 and max_rel(pyg.cpu().numpy(), o_gy) <= 1e-4
    loss.sum().backward()       # gradients reach am and lm through the native op
    assert torch.isfinite(am.grad).all() and torch.isfinite(lm.grad).all()
    # d loss / d am summed over classes is 0 (the model is normalised over C for every (s,t))
    np.testing.assert_allclose(am.grad.sum(dim=2).cpu().numpy(), 0.0, atol=2e-4)


def test_mean_reduction_and_defaults(ft, dev, oracle):
    d = synthetic(2, 2, 8, 4, 16)           # BASELINE config c1 shape: B=2 T=8 S=4 C=16
    out = ft.rnnt_loss_simple(_t(d["lm"], dev), _t(d["am"], dev), _t(d["symbols"], dev), d["termination_symbol"])
    want = oracle.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], d["termination_symbol"])
    np.testing.assert_allclose(out.item(), want, rtol=1e-5)
    with pytest.raises(ValueError):
        ft.rnnt_loss_simple(_t(d["lm"], dev), _t(d["am"], dev), _t(d["symbols"], dev), d["termination_symbol"], reduction="bogus")


def test_full_size_pruned_step_properties(ft, dev):
    """BASELINE config c3 (B=32,T=1000,S=200,C=500,s_range=5) end to end on the GPU; checks the
    size-independent properties: ranges monotone / bounded / pinned at the last frame, gather idempotent,
    pruned loss finite and >= the unpruned-path lower bound property loss_pruned >= simple-lattice bound is
    not defined, so: gradient rows sum to ~0 (softmax gradient) and occupancies sum to 1 per frame."""
    from bench import make_inputs, pruned_step
    inp = make_inputs(B=32, T=1000, S=200, C=500, seed=0, device=dev)
    out = pruned_step(inp, s_range=5, keep=True)
    torch.cuda.synchronize()
    ranges = out["ranges"].cpu().numpy()
    s0 = ranges[:, :, 0]
    assert (np.diff(s0, axis=1) >= 0).all() and (np.diff(s0, axis=1) <= 4).all() and (s0 >= 0).all()
    assert (s0[:, -1] == 200 - 5 + 1).all()
    assert (ranges == s0[:, :, None] + np.arange(5)).all()
    pyg = out["py_grad"]
    np.testing.assert_allclose(pyg.sum(dim=1).cpu().numpy(), 1.0, rtol=1e-4)
    g = out["logits_grad"]
    assert torch.isfinite(g).all() and torch.isfinite(out["pruned_loss"]).all()
    np.testing.assert_allclose(g.sum(dim=3).cpu().numpy(), 0.0, atol=1e-5)


def test_full_size_smoothed_pruned_step_properties(ft, dev):
    """BASELINE config c4's per-GPU share (B=32, T=2000, S=300, C=1024, s_range=5) with the smoothed first pass, ragged
    boundaries: the same size-independent properties as above (ranges monotone with steps <= s_range-1, occupancies sum
    to 1 per valid frame, softmax-gradient rows sum to 0), plus finite gradients w.r.t. am and lm."""
    from bench import make_inputs, pruned_step
    inp = make_inputs(B=32, T=2000, S=300, C=1024, seed=1, device=dev, ragged=True)
    out = pruned_step(inp, s_range=5, keep=True, first_pass="smoothed")
    torch.cuda.synchronize()
    s0 = out["ranges"].cpu().numpy()[:, :, 0]
    assert (np.diff(s0, axis=1) >= 0).all() and (np.diff(s0, axis=1) <= 4).all() and (s0 >= 0).all() and (s0 <= 300 - 5 + 1).all()
    bd = inp["boundary"].cpu().numpy()
    pyg = out["py_grad"].sum(dim=1).cpu().numpy()
    for b in range(32):
        np.testing.assert_allclose(pyg[b, :bd[b, 3]], 1.0, rtol=2e-4)
        assert not pyg[b, bd[b, 3]:].any()
    g = out["logits_grad"]
    assert torch.isfinite(g).all() and torch.isfinite(out["pruned_loss"]).all() and torch.isfinite(out["simple_loss"]).all()
    np.testing.assert_allclose(g.sum(dim=3).cpu().numpy(), 0.0, atol=2e-5)
    assert torch.isfinite(out["am_grad"]).all() and torch.isfinite(out["lm_grad"]).all()


@pytest.mark.parametrize("rnnt_type", ["regular", "modified", "constrained"])
@pytest.mark.parametrize("cfg", [(3, 24, 8, 12), (2, 70, 33, 50), (2, 33, 5, 7), (2, 37, 9, 700), (1, 21, 4, 641)])
def test_native_simple_builder_forward_backward(ft, dev, oracle, rnnt_type, cfg):
    """get_rnnt_logprobs (native prologue/epilogue kernels around the GEMM): px/py against the oracle with the exact
    -inf pattern; d/d am and d/d lm against float64 autograd through the op-by-op torch restatement."""
    from torch_restatements import get_rnnt_logprobs_torch as _get_rnnt_logprobs_torch
    B, T, S, C = cfg
    d = synthetic(11, B, T, S, C, ragged=True)
    am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
    sym = _t(d["symbols"], dev); bd = _t(d["boundary"], dev)
    px, py = ft.get_rnnt_logprobs(lm, am, sym, d["termination_symbol"], rnnt_type, bd)
    o_px, o_py = oracle.get_rnnt_logprobs(d["lm"], d["am"], d["symbols"], d["termination_symbol"], rnnt_type, d["boundary"])
    pxn = px.detach().cpu().numpy(); pyn = py.detach().cpu().numpy()
    assert pxn.shape == o_px.shape and pyn.shape == o_py.shape
    assert np.array_equal(np.isneginf(pxn), np.isneginf(o_px))
    fin = np.isfinite(o_px)
    np.testing.assert_allclose(pxn[fin], o_px[fin], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pyn, o_py, rtol=1e-5, atol=2e-5)
    g = torch.Generator(device="cpu").manual_seed(3)
    wx = torch.randn(px.shape, generator=g).to(dev); wy = torch.randn(py.shape, generator=g).to(dev)
    finite = torch.isfinite(px.detach())
    (torch.where(finite, px, torch.zeros_like(px)) * wx).sum().add((py * wy).sum()).backward()
    am64 = am.detach().double().requires_grad_(True); lm64 = lm.detach().double().requires_grad_(True)
    px64, py64 = _get_rnnt_logprobs_torch(lm64, am64, sym, d["termination_symbol"], rnnt_type, bd)
    (torch.where(finite, px64, torch.zeros_like(px64)) * wx.double()).sum().add((py64 * wy.double()).sum()).backward()
    assert max_rel(am.grad.cpu().numpy(), am64.grad.cpu().numpy()) <= 1e-4
    assert max_rel(lm.grad.cpu().numpy(), lm64.grad.cpu().numpy()) <= 1e-4


def test_native_simple_builder_odd_vocab_and_penalty(ft, dev, oracle):
    """C not a multiple of 4 (scalar row path) and the fused delay penalty of rnnt_loss_simple."""
    d = synthetic(12, 2, 19, 6, 11, ragged=True)
    args = (_t(d["lm"], dev), _t(d["am"], dev), _t(d["symbols"], dev), d["termination_symbol"])
    for rt in ("regular", "modified", "constrained"):
        got = ft.rnnt_loss_simple(*args, boundary=_t(d["boundary"], dev), rnnt_type=rt, delay_penalty=0.3, reduction="none")
        want = oracle.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], d["termination_symbol"], d["boundary"], rt, 0.3, "none")
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ft.get_rnnt_logprobs(_t(d["lm"], "cpu"), _t(d["am"], "cpu"), _t(d["symbols"], "cpu"), d["termination_symbol"])


@pytest.mark.parametrize("rnnt_type", ["regular", "modified", "constrained"])
@pytest.mark.parametrize("scales", [(0.1, 0.2), (0.0, 0.0), (0.25, 0.0)])
@pytest.mark.parametrize("cfg", [(3, 24, 8, 12), (2, 70, 33, 50), (2, 33, 5, 7), (2, 37, 9, 644)])
def test_native_smoothed_builder_forward_backward(ft, dev, oracle, rnnt_type, scales, cfg):
    """get_rnnt_logprobs_smoothed on the native builder kernels (rnnt_loss.py:1132-1367): px/py against the oracle
    with the exact -inf pattern (tolerance 2e-5 absolute/relative: f32 sums in a different order); d/d am and d/d lm
    against float64 autograd through the op-by-op torch restatement, 1e-4 normwise."""
    from torch_restatements import get_rnnt_logprobs_smoothed_torch as _get_rnnt_logprobs_smoothed_torch
    B, T, S, C = cfg
    d = synthetic(21, B, T, S, C, ragged=True)
    am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
    sym = _t(d["symbols"], dev); bd = _t(d["boundary"], dev)
    px, py = ft.get_rnnt_logprobs_smoothed(lm, am, sym, d["termination_symbol"], scales[0], scales[1], bd, rnnt_type)
    o_px, o_py = oracle.get_rnnt_logprobs_smoothed(d["lm"], d["am"], d["symbols"], d["termination_symbol"],
                                                   scales[0], scales[1], d["boundary"], rnnt_type)
    pxn = px.detach().cpu().numpy(); pyn = py.detach().cpu().numpy()
    assert pxn.shape == o_px.shape and pyn.shape == o_py.shape
    assert np.array_equal(np.isneginf(pxn), np.isneginf(o_px))
    fin = np.isfinite(o_px)
    np.testing.assert_allclose(pxn[fin], o_px[fin], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(pyn, o_py, rtol=2e-5, atol=2e-5)
    g = torch.Generator(device="cpu").manual_seed(3)
    wx = torch.randn(px.shape, generator=g).to(dev); wy = torch.randn(py.shape, generator=g).to(dev)
    finite = torch.isfinite(px.detach())
    (torch.where(finite, px, torch.zeros_like(px)) * wx).sum().add((py * wy).sum()).backward()
    am64 = am.detach().double().requires_grad_(True); lm64 = lm.detach().double().requires_grad_(True)
    px64, py64 = _get_rnnt_logprobs_smoothed_torch(lm64, am64, sym, d["termination_symbol"], scales[0], scales[1], bd,
                                                   rnnt_type)
    (torch.where(finite, px64, torch.zeros_like(px64)) * wx.double()).sum().add((py64 * wy.double()).sum()).backward()
    assert max_rel(am.grad.cpu().numpy(), am64.grad.cpu().numpy()) <= 1e-4
    assert max_rel(lm.grad.cpu().numpy(), lm64.grad.cpu().numpy()) <= 1e-4


def test_smoothed_loss_against_oracle(ft, dev, oracle):
    """rnnt_loss_smoothed end to end (simple_rnnt_loss_test.py:291-336 scales and penalty): loss and occupancies."""
    d = reference_test_recipe(12345, 2, 200, 50, 50)
    args = (_t(d["lm"], dev), _t(d["am"], dev), _t(d["symbols"], dev), d["termination_symbol"])
    for rt in ("regular", "modified"):
        got, (gx, gy) = ft.rnnt_loss_smoothed(*args, lm_only_scale=0.1, am_only_scale=0.2, boundary=_t(d["boundary"], dev),
                                              rnnt_type=rt, delay_penalty=0.2, reduction="none", calc_gradients=True)
        want, (ox, oy) = oracle.rnnt_loss_smoothed(d["lm"], d["am"], d["symbols"], d["termination_symbol"], 0.1, 0.2,
                                                   d["boundary"], rt, 0.2, "none", True)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4)
        # occupancies against the float64 recursion on the oracle's float32 px / py (the float32 recursion is 3e-4 off at T=200)
        opx, opy = oracle.get_rnnt_logprobs_smoothed(d["lm"], d["am"], d["symbols"], d["termination_symbol"], 0.1, 0.2, d["boundary"], rt)
        opx = oracle._delay_penalty(opx, d["boundary"], rt, 0.2)
        _, (ox64, oy64) = oracle.mutual_information_recursion(opx, opy, d["boundary"], True, np.float64)
        assert max_rel(gx.cpu().numpy(), ox64) <= 2e-5 and max_rel(gy.cpu().numpy(), oy64) <= 2e-5
        assert max_rel(gx.cpu().numpy(), ox) <= 1.05 * max_rel(ox, ox64) + 2e-5 and max_rel(gy.cpu().numpy(), oy) <= 1.05 * max_rel(oy, oy64) + 2e-5


@pytest.mark.parametrize("rnnt_type", ["regular", "modified", "constrained"])
def test_native_joint_builder_and_unpruned_loss(ft, dev, oracle, rnnt_type):
    """get_rnnt_logprobs_joint / rnnt_loss (rnnt_loss.py:340-551) on the pruned builder's kernels with identity ranges:
    px/py against the oracle, loss against the oracle, d/d logits against float64 autograd of the torch restatement."""
    from torch_restatements import get_rnnt_logprobs_joint_torch as _get_rnnt_logprobs_joint_torch
    d = synthetic(31, 2, 21, 6, 9, ragged=True)
    logits_np = (d["am"][:, :, None, :] + d["lm"][:, None, :, :]).astype(np.float32)
    logits = _t(logits_np, dev).requires_grad_(True)
    sym = _t(d["symbols"], dev); bd = _t(d["boundary"], dev); blank = d["termination_symbol"]
    px, py = ft.get_rnnt_logprobs_joint(logits, sym, blank, bd, rnnt_type)
    o_px, o_py = oracle.get_rnnt_logprobs_joint(logits_np, d["symbols"], blank, d["boundary"], rnnt_type)
    pxn = px.detach().cpu().numpy()
    assert np.array_equal(np.isneginf(pxn), np.isneginf(o_px))
    fin = np.isfinite(o_px)
    np.testing.assert_allclose(pxn[fin], o_px[fin], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(py.detach().cpu().numpy(), o_py, rtol=1e-5, atol=2e-5)
    loss = ft.rnnt_loss(logits, sym, blank, bd, rnnt_type, delay_penalty=0.1, reduction="sum")
    want = oracle.rnnt_loss(logits_np, d["symbols"], blank, d["boundary"], rnnt_type, 0.1, "sum")
    np.testing.assert_allclose(loss.item(), want, rtol=1e-4)
    loss.backward()
    l64 = logits.detach().double().requires_grad_(True)
    px64, py64 = _get_rnnt_logprobs_joint_torch(l64, sym, blank, bd, rnnt_type)
    from tf_fast_rnnt.rnnt_loss import _apply_delay_penalty
    px64 = _apply_delay_penalty(px64, bd, rnnt_type, 0.1)
    # float64 DP in torch (log-domain, autograd) as the gradient reference
    B, S, T1 = px64.shape; T = py64.shape[2]
    tot = 0.0
    for b in range(B):
        sb, tb, se, te = [int(v) for v in d["boundary"][b]]
        p = {}
        for s in range(sb, se + 1):
            for t in range(tb, te + 1):
                if s == sb and t == tb:
                    p[(s, t)] = px64.new_zeros(()); continue
                terms = []
                if s > sb:
                    tt = t if rnnt_type == "regular" else t - 1
                    if tt >= tb and (s - 1, tt) in p and torch.isfinite(px64[b, s - 1, tt]):
                        terms.append(p[(s - 1, tt)] + px64[b, s - 1, tt])
                if t > tb and (s, t - 1) in p:
                    terms.append(p[(s, t - 1)] + py64[b, s, t - 1])
                if terms:
                    p[(s, t)] = torch.logsumexp(torch.stack(terms), 0)
        tot = tot - p[(se, te)]
    tot.backward()
    assert abs(tot.item() - loss.item()) <= 1e-4 * abs(tot.item())
    assert max_rel(logits.grad.cpu().numpy(), l64.grad.cpu().numpy()) <= 1e-4


@pytest.mark.parametrize("same_tensor", [True, False])
@pytest.mark.parametrize("kind", ["monotone", "scattered", "wide"])
def test_do_pruning_backward_chunked(ft, dev, kind, same_tensor):
    """The segmented (workspace) backward of the prune gather on caller-chosen ranges: monotone bands (window path),
    rows scattered over the whole lattice and bands that jump back (both take the rescan path of pass 2), with the two
    incoming gradients being one tensor (fused d_am) or two.  float64 segment sums as reference;
    tolerance 1e-5 relative (f32 sums in a different association)."""
    B, T, S1, C, r = 3, 70, 40, 24, 5
    g = torch.Generator(device="cpu").manual_seed(17)
    if kind == "monotone":
        s0 = torch.sort(torch.randint(0, S1 - r + 1, (B, T), generator=g), dim=1).values
        ranges = s0.unsqueeze(2) + torch.arange(r)
    elif kind == "scattered":
        ranges = torch.randint(0, S1, (B, T, r), generator=g)
    else:   # consecutive frames jump by more than the bin budget
        s0 = (torch.arange(T) * 7) % (S1 - r + 1)
        ranges = (s0.unsqueeze(1) + torch.arange(r)).expand(B, T, r).clone()
    ranges = ranges.to(torch.int32).to(dev)
    am = torch.randn((B, T, C), generator=g).to(dev).requires_grad_(True)
    lm = torch.randn((B, S1, C), generator=g).to(dev).requires_grad_(True)
    am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
    w = torch.randn((B, T, r, C), generator=g).to(dev)
    if same_tensor:
        ((am_p + lm_p) * w).sum().backward()          # autograd hands one buffer to both inputs
        w2 = w
    else:
        w2 = torch.randn((B, T, r, C), generator=g).to(dev)
        ((am_p * w).sum() + (lm_p * w2).sum()).backward()
    np.testing.assert_allclose(am.grad.cpu().numpy(), w.double().sum(dim=2).cpu().numpy(), rtol=1e-5, atol=1e-6)
    want = torch.zeros((B, S1, C), dtype=torch.float64, device=dev)
    want.index_put_((torch.arange(B, device=dev).view(B, 1, 1).expand(B, T, r), ranges.long()), w2.double(), accumulate=True)
    np.testing.assert_allclose(lm.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind", ["flat_then_steep", "gaps", "half_garbage", "owner_conflict", "out_of_range"])
@pytest.mark.parametrize("shape", [(3, 150, 60, 24, 5), (2, 90, 50, 520, 3), (2, 64, 80, 40, 10), (1, 50, 90, 16, 17)])
def test_do_pruning_backward_segments(ft, dev, kind, shape, monkeypatch):
    """The segmented backward of the prune gather (csrc/prune.hip): rows a segment owns go straight to d lm, rows shared
    with a neighbour through partial rows, everything that is not a band through the rescan of pass 2.  Shapes: several
    segments, more than 128 column quads (two sweeps), windows of 16 and 32 rows; 16-frame segments forced so that the
    small T still has many.  float64 segment sums as reference."""
    monkeypatch.setenv("FTR_PRUNE_SEG", "16")
    B, T, S1, C, r = shape
    g = torch.Generator(device="cpu").manual_seed(23)
    top = S1 - r
    if kind == "flat_then_steep":        # what random occupancies give: the same rows for most frames, then r-1 rows per frame
        s0 = torch.zeros((B, T), dtype=torch.int64)
        climb = torch.clamp((torch.arange(T) - (T - top // max(r - 1, 1) - 3)) * max(r - 1, 1), 0, top)
        s0 += climb
    elif kind == "gaps":                 # monotone, but frames jump over rows nobody touches
        s0 = torch.clamp(torch.cumsum((torch.rand((B, T), generator=g) < 0.1).long() * (r + 3), 1), 0, top)
    elif kind == "half_garbage":         # a band in the first half of the frames, arbitrary rows in the second
        s0 = torch.sort(torch.randint(0, top + 1, (B, T), generator=g), dim=1).values
    elif kind == "owner_conflict":       # every segment is a band, but a later one returns to rows an earlier one owns
        s0 = torch.sort(torch.randint(0, top + 1, (B, T), generator=g), dim=1).values
        s0[:, T // 2:] = torch.sort(torch.randint(0, top + 1, (B, T - T // 2), generator=g), dim=1).values
    else:                                # rows outside [0, S1): ignored, must not fault
        s0 = torch.sort(torch.randint(0, top + 1, (B, T), generator=g), dim=1).values
    ranges = (s0.unsqueeze(2) + torch.arange(r)).clone()
    if kind == "half_garbage":
        ranges[:, T // 2:] = torch.randint(0, S1, (B, T - T // 2, r), generator=g)
    ranges = ranges.to(torch.int32).to(dev)
    w = torch.randn((B, T, r, C), generator=g).to(dev)
    idx = ranges.long()
    if kind == "out_of_range":
        ranges = ranges.clone(); ranges[:, 5:9] += S1; ranges[:, 20:22] -= 2 * S1
        idx = ranges.long()
    d_am = torch.full((B, T, C), float("nan"), device=dev); d_lm = torch.full((B, S1, C), float("nan"), device=dev)
    L = ft._lib
    nbytes = int(L.lib().ftr_do_pruning_bwd_workspace_bytes(B, T, S1, C, r))
    ws = torch.empty((nbytes + 3) // 4, device=dev)
    L.call("ftr_do_pruning_bwd_ws_f32", w.data_ptr(), w.data_ptr(), ranges.data_ptr(), d_am.data_ptr(), d_lm.data_ptr(),
           B, T, S1, C, r, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    np.testing.assert_allclose(d_am.cpu().numpy(), w.double().sum(dim=2).cpu().numpy(), rtol=1e-5, atol=1e-5)
    want = torch.zeros((B, S1, C), dtype=torch.float64, device=dev)
    valid = (idx >= 0) & (idx < S1)
    bi = torch.arange(B, device=dev).view(B, 1, 1).expand(B, T, r)
    want.index_put_((bi[valid], idx[valid]), w.double()[valid], accumulate=True)
    np.testing.assert_allclose(d_lm.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("reduction", ["none", "mean", "sum"])
@pytest.mark.parametrize("rnnt_type", ["regular", "modified"])
def test_fused_loss_nodes_match_composed_path(ft, dev, reduction, rnnt_type):
    """rnnt_loss_simple / rnnt_loss_pruned as single autograd nodes (native reduction, upstream gradient folded into the
    backward kernels: ftr_negated_reduce_f32, ftr_*_scaled_f32) against the same losses composed from the public
    pieces (px/py builder -> mutual_information_recursion -> torch reduction), values and gradients, with a non-trivial
    upstream gradient.  Same kernels underneath: 1e-5."""
    d = synthetic(41, 3, 30, 9, 12, ragged=True)
    sym = _t(d["symbols"], dev); bd = _t(d["boundary"], dev); blank = d["termination_symbol"]
    g = torch.Generator(device="cpu").manual_seed(9)
    wgt = torch.rand((3,), generator=g).to(dev) + 0.5

    def run(fused):
        am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
        if fused:
            loss, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, blank, bd, rnnt_type, 0.15, reduction, True)
        else:
            px, py = ft.get_rnnt_logprobs(lm, am, sym, blank, rnnt_type, bd)
            from tf_fast_rnnt.rnnt_loss import _apply_delay_penalty, _reduce
            px = _apply_delay_penalty(px, bd, rnnt_type, 0.15)
            ans, (gx, gy) = ft.mutual_information_recursion(px, py, bd, True)
            loss = _reduce(ans, reduction)
        ranges = ft.get_rnnt_prune_ranges(gx, gy, bd, 4)
        am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
        logits = torch.tanh(am_p + lm_p)
        if fused:
            ploss = ft.rnnt_loss_pruned(logits, sym, ranges, blank, bd, rnnt_type, 0.1, reduction)
        else:
            from tf_fast_rnnt.rnnt_loss import _apply_delay_penalty, _reduce
            ppx, ppy = ft.get_rnnt_logprobs_pruned(logits, sym, ranges, blank, bd, rnnt_type)
            ppx = _apply_delay_penalty(ppx, bd, rnnt_type, 0.1)
            ploss = _reduce(ft.mutual_information_recursion(ppx, ppy, bd), reduction)
        total = (loss * wgt).sum() + 0.7 * (ploss * wgt).sum() if reduction == "none" else 1.3 * loss + 0.7 * ploss
        total.backward()
        return loss.detach().cpu().numpy(), ploss.detach().cpu().numpy(), am.grad.cpu().numpy(), lm.grad.cpu().numpy()

    a = run(True); b = run(False)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5)
    assert max_rel(a[2], b[2]) <= 1e-5 and max_rel(a[3], b[3]) <= 1e-5


@pytest.mark.parametrize("reduction", ["none", "mean", "sum"])
@pytest.mark.parametrize("rnnt_type", ["regular", "modified"])
def test_fused_smoothed_loss_matches_composed_path(ft, dev, reduction, rnnt_type):
    """rnnt_loss_smoothed as a single autograd node (penalty inside the builder kernel, ftr_smoothed_logprobs_bwd_*_scaled)
    against the same loss composed from get_rnnt_logprobs_smoothed -> penalty -> mutual_information_recursion -> torch
    reduction: value, occupancies and d/d am, d/d lm with a non-trivial upstream gradient.  Same kernels: 1e-5."""
    from tf_fast_rnnt.rnnt_loss import _apply_delay_penalty, _reduce
    d = synthetic(43, 3, 31, 8, 14, ragged=True)
    sym = _t(d["symbols"], dev); bd = _t(d["boundary"], dev); blank = d["termination_symbol"]
    wgt = torch.rand((3,), generator=torch.Generator(device="cpu").manual_seed(5)).to(dev) + 0.5

    def run(fused):
        am = _t(d["am"], dev).requires_grad_(True); lm = _t(d["lm"], dev).requires_grad_(True)
        if fused:
            loss, (gx, gy) = ft.rnnt_loss_smoothed(lm, am, sym, blank, 0.1, 0.2, bd, rnnt_type, 0.15, reduction, True)
        else:
            px, py = ft.get_rnnt_logprobs_smoothed(lm, am, sym, blank, 0.1, 0.2, bd, rnnt_type)
            px = _apply_delay_penalty(px, bd, rnnt_type, 0.15)
            ans, (gx, gy) = ft.mutual_information_recursion(px, py, bd, True)
            loss = _reduce(ans, reduction)
        total = (loss * wgt).sum() if reduction == "none" else 1.7 * loss
        total.backward()
        return loss.detach().cpu().numpy(), gx.cpu().numpy(), gy.cpu().numpy(), am.grad.cpu().numpy(), lm.grad.cpu().numpy()

    a = run(True); b = run(False)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-5)
    for i in (1, 2, 3, 4):
        assert max_rel(a[i], b[i]) <= 1e-5, i
    # no occupancies wanted, no autograd: the recursion backward is skipped and zeros are returned
    with torch.no_grad():
        only = ft.rnnt_loss_smoothed(_t(d["lm"], dev), _t(d["am"], dev), sym, blank, 0.1, 0.2, bd, rnnt_type, 0.15, reduction)
    np.testing.assert_allclose(only.cpu().numpy(), a[0], rtol=1e-6)


@pytest.mark.parametrize("rnnt_type", ["regular", "modified"])
@pytest.mark.parametrize("cfg", [(3, 40, 12, 20, 4), (2, 90, 33, 12, 5), (4, 64, 20, 16, 2), (2, 130, 50, 24, 8), (2, 70, 40, 8, 16), (2, 70, 40, 8, 15), (2, 60, 30, 8, 7),
                                 (3, 33, 5, 7, 3), (2, 200, 50, 50, 5), (1, 300, 10, 16, 11), (2, 25, 20, 8, 6)])
def test_band_native_pruned_loss_matches_lattice_path(ft, dev, oracle, rnnt_type, cfg, monkeypatch):
    """rnnt_loss_pruned on the band itself (ftr_mutual_information_band_f32) against the same loss through full-size
    lattices (FTR_PRUNED_ROUTE=lattice): loss and d/d logits, ragged
    boundaries, delay penalty, non-trivial upstream gradient.  Two float32 evaluations of the same quantity: 1e-4 on the
    loss, 2e-4 normwise on the gradient (observed ~1e-6 / ~1e-5); and both against the float64 oracle."""
    B, T, S, C, r = cfg
    d = synthetic(100 + T + S, B, T, S, C, ragged=True)
    blank = d["termination_symbol"]
    am, lm, sym, bd = (_t(d[k], dev) for k in ("am", "lm", "symbols", "boundary"))
    _, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, blank, bd, rnnt_type, reduction="sum", calc_gradients=True)
    ranges = ft.get_rnnt_prune_ranges(gx, gy, bd, r)
    am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
    base = torch.tanh(am_p + lm_p).detach()
    wgt = torch.rand((B,), generator=torch.Generator(device="cpu").manual_seed(1)).to(dev) + 0.5
    outs = []
    for route in ("band", "lattice"):
        monkeypatch.setenv("FTR_PRUNED_ROUTE", route)
        logits = base.clone().requires_grad_(True)
        loss = ft.rnnt_loss_pruned(logits, sym, ranges, blank, bd, rnnt_type, 0.1, "none")
        (loss * wgt).sum().backward()
        outs.append((loss.detach().cpu().numpy(), logits.grad.cpu().numpy()))
    fin = np.isfinite(outs[1][0])
    assert np.array_equal(np.isfinite(outs[0][0]), fin)
    np.testing.assert_allclose(outs[0][0][fin], outs[1][0][fin], rtol=1e-4)
    assert max_rel(outs[0][1][fin], outs[1][1][fin]) <= 2e-4
    if rnnt_type == "regular":
        o_loss, o_g = oracle.rnnt_loss_pruned_grad(base.cpu().numpy(), d["symbols"], ranges.cpu().numpy(), blank, d["boundary"],
                                                   delay_penalty=0.1, reduction="none", dtype=np.float64)
        np.testing.assert_allclose(outs[0][0][fin], o_loss[fin], rtol=1e-4)
        assert max_rel(outs[0][1][fin], (o_g * wgt.cpu().numpy().reshape(-1, 1, 1, 1))[fin]) <= 2e-4


def test_pruned_loss_routes_by_the_data_not_by_a_mark(ft, dev, monkeypatch):
    """A clone / a slice / a reloaded copy of get_rnnt_prune_ranges' output carries no mark and must still take the
    band-native route (one device-side check, remembered on the tensor); a ranges tensor that is NOT a band (non-monotone,
    or rows that are not consecutive) must take the lattice route and give the lattice route's answer; an in-place edit
    invalidates what was remembered."""
    import io
    from tf_fast_rnnt import _lib
    from tf_fast_rnnt.rnnt_loss import _is_band
    B, T, S, C, r = 3, 90, 33, 12, 5
    d = synthetic(321, B, T, S, C, ragged=True)
    blank = d["termination_symbol"]
    am, lm, sym, bd = (_t(d[k], dev) for k in ("am", "lm", "symbols", "boundary"))
    _, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, blank, bd, reduction="sum", calc_gradients=True)
    ranges = ft.get_rnnt_prune_ranges(gx, gy, bd, r)
    am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
    logits = torch.tanh(am_p + lm_p).detach()
    calls = []
    real_call = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real_call(name, *a))[1])
    buf = io.BytesIO(); torch.save(ranges.cpu(), buf); buf.seek(0)
    want = ft.rnnt_loss_pruned(logits, sym, ranges, blank, bd, reduction="none").cpu().numpy()
    assert "ftr_mutual_information_band_ws_f32" in calls and "ftr_band_ranges_check_i32" not in calls    # marked: no check
    for other in (ranges.clone(), ranges[:, :, :].contiguous(), torch.load(buf, weights_only=True).to(dev), ranges.to(torch.int64)):
        calls.clear()
        got = ft.rnnt_loss_pruned(logits, sym, other, blank, bd, reduction="none").cpu().numpy()
        assert "ftr_mutual_information_band_ws_f32" in calls and "ftr_pruned_logprobs_fwd_f32" not in calls
        assert np.array_equal(got, want)                      # the same kernels on the same data
    c = ranges.clone()
    calls.clear(); ft.rnnt_loss_pruned(logits, sym, c, blank, bd, reduction="none")
    assert calls.count("ftr_band_ranges_check_i32") == 1
    calls.clear(); ft.rnnt_loss_pruned(logits, sym, c, blank, bd, reduction="none")
    assert "ftr_band_ranges_check_i32" not in calls           # the verdict is remembered
    # not a band: one frame steps backwards (utterance 1), or a row is not consecutive (utterance 2)
    for edit in ("backwards", "gap"):
        bad = ranges.clone()
        assert _is_band(bad, bd)
        t0 = int(bd[1, 3].item()) // 2
        if edit == "backwards":
            bad[1, t0] = torch.clamp(bad[1, t0 - 1] - 1, min=0)      # in-place: the remembered verdict must not survive
        else:
            bad[2, 3, r - 1] = bad[2, 3, r - 1] - 1
        if edit == "backwards" and int(bad[1, t0, 0]) >= int(bad[1, t0 - 1, 0]):
            continue                                                 # already at row 0: nothing to break
        assert not _is_band(bad, bd)
        calls.clear()
        got = ft.rnnt_loss_pruned(logits, sym, bad, blank, bd, reduction="none").cpu().numpy()
        assert "ftr_pruned_logprobs_fwd_f32" in calls and "ftr_mutual_information_band_ws_f32" not in calls
        monkeypatch.setenv("FTR_PRUNED_ROUTE", "lattice")
        ref = ft.rnnt_loss_pruned(logits, sym, bad.clone(), blank, bd, reduction="none").cpu().numpy()
        monkeypatch.delenv("FTR_PRUNED_ROUTE")
        assert np.array_equal(got, ref, equal_nan=True)


def test_out_of_range_caller_data_does_not_fault(ft, dev):
    """Malformed boundary rows, symbols and ranges (caller data the reference never validates) must not send a kernel
    out of bounds: boundaries are clamped into the lattice, symbols into the vocabulary, gather rows into lm; rows of
    `ranges` outside the lattice receive no gradient.  Only "runs, finite where defined" is asserted."""
    B, T, S, C, r = 2, 20, 6, 8, 3
    g = torch.Generator(device="cpu").manual_seed(3)
    am = torch.randn((B, T, C), generator=g).to(dev).requires_grad_(True)
    lm = torch.randn((B, S + 1, C), generator=g).to(dev).requires_grad_(True)
    sym = torch.tensor([[0, 1, 99, -5, 2, 3], [7, 7, 7, 1000000, 0, 1]], dtype=torch.int32, device=dev)
    bd = torch.tensor([[0, 0, S + 50, T + 70], [-3, -2, S, T]], dtype=torch.int32, device=dev)
    loss, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, C - 1, bd, reduction="sum", calc_gradients=True)
    ranges = torch.tensor([-4, 0, 1000], dtype=torch.int32, device=dev).expand(B, T, r).contiguous()
    am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
    ploss = ft.rnnt_loss_pruned(torch.tanh(am_p + lm_p), sym, ranges, C - 1, bd, reduction="sum")
    (loss + ploss).backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and torch.isfinite(am.grad).all() and torch.isfinite(lm.grad).all()


@pytest.mark.parametrize("name", ["c1_B2_T8_S4_C16", "seed1234_B2_T10_S7_C4", "seed12345_B2_T200_S50_C50"])
def test_against_committed_golden_fixtures(ft, dev, name):
    """The HIP path against the committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
    oracle; the reference holds no expected values of its own): every stage of the pipeline on the fixture's inputs.
    Ranges bit-exact given the fixture's occupancies; losses 1e-4 elementwise; occupancies and gradients 2e-5 normwise
    against the fixture's *_f64 arrays (the same float32 px / py through the recursion in float64; measured 1e-6 ... 8e-6),
    and against the float32 arrays (the reference's arithmetic) within that arithmetic's own distance from the float64
    arrays, which is 3e-4 at T=200."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    blank = int(g["termination_symbol"])
    am, lm, sym, bd = (_t(g[k], dev) for k in ("am", "lm", "symbols", "boundary"))

    def close(got, key):
        got = got.cpu().numpy()
        assert max_rel(got, g[key + "_f64"]) <= 2e-5, (name, key, max_rel(got, g[key + "_f64"]))
        assert max_rel(got, g[key]) <= max(1e-5, 1.05 * max_rel(g[key], g[key + "_f64"])) + 2e-5, (name, key)
    px, py = ft.get_rnnt_logprobs(lm, am, sym, blank, "regular", bd)
    assert np.array_equal(np.isneginf(px.cpu().numpy()), np.isneginf(g["simple_px"]))
    assert max_rel(px.cpu().numpy(), g["simple_px"]) <= 1e-5 and max_rel(py.cpu().numpy(), g["simple_py"]) <= 1e-5
    loss, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, blank, bd, reduction="none", calc_gradients=True)
    np.testing.assert_allclose(loss.cpu().numpy(), g["simple_loss"], rtol=1e-4)
    close(gx, "simple_px_grad"); close(gy, "simple_py_grad")
    sl, (sgx, sgy) = ft.rnnt_loss_smoothed(lm, am, sym, blank, lm_only_scale=0.1, am_only_scale=0.2, boundary=bd,
                                           reduction="none", delay_penalty=0.2, calc_gradients=True)
    np.testing.assert_allclose(sl.cpu().numpy(), g["smoothed_loss"], rtol=1e-4)
    close(sgx, "smoothed_px_grad"); close(sgy, "smoothed_py_grad")
    for r in [int(v) for v in g["s_ranges"]]:
        want = g[f"ranges_r{r}"]
        got = ft.get_rnnt_prune_ranges(_t(g["smoothed_px_grad"], dev), _t(g["smoothed_py_grad"], dev), bd, r)
        assert np.array_equal(got.cpu().numpy(), want)                      # integer output: bit-exact
        am_p, lm_p = ft.do_rnnt_pruning(am, lm, _t(want, dev))
        logits = torch.sigmoid(am_p + lm_p).detach().requires_grad_(True)
        pl = ft.rnnt_loss_pruned(logits, sym, _t(want, dev), blank, bd, delay_penalty=0.2, reduction="mean")
        pl.backward()
        np.testing.assert_allclose(pl.item(), float(g[f"pruned_loss_r{r}"]), rtol=1e-4)
        close(logits.grad, f"pruned_logits_grad_r{r}")


@pytest.mark.parametrize("blocks", ["4", "7", "10", "13"])
@pytest.mark.parametrize("frames", ["64", "128"])
@pytest.mark.parametrize("rnnt_type", ["regular", "modified"])
@pytest.mark.parametrize("cfg", [(2, 70, 33, 12), (2, 129, 100, 20), (1, 200, 140, 16), (2, 65, 200, 8), (1, 63, 470, 24), (3, 64, 15, 36),
                                 (2, 1, 0, 4), (2, 5, 0, 8), (1, 3, 70, 4), (2, 64, 1, 4), (1, 300, 40, 72)])
def test_fused_builder_matches_library_gemm_route(ft, dev, oracle, rnnt_type, cfg, frames, blocks, monkeypatch):
    """csrc/simple_fused.hip (f32-MFMA contraction + epilogue in one kernel) against the library-GEMM + epilogue route with
    every symbol-block count (4 / 7 / 10 / 13 per workgroup: the launcher would take 4 for problems this small, so the count
    is forced), one and several symbol tiles (with several, the tile order that puts a frame block's symbol tiles side by
    side), ragged frame tiles, boundaries and the penalty; simple and smoothed; with 64- and with 128-frame tiles (the launcher
    takes the latter only for large lm_probs row sets: forced here).  Same -inf pattern, values to 2e-5 (summation order
    differs)."""
    monkeypatch.setenv("FTR_FUSED_FT", frames)
    monkeypatch.setenv("FTR_FUSED_NS", blocks)
    B, T, S, C = cfg
    d = synthetic(11 + S, B, T, S, C, ragged=True)
    lm, am, sym, bnd = (_t(d[k], dev) for k in ("lm", "am", "symbols", "boundary"))
    blank = d["termination_symbol"]

    def run():
        a = ft.get_rnnt_logprobs(lm, am, sym, blank, rnnt_type=rnnt_type, boundary=bnd)
        b = ft.get_rnnt_logprobs_smoothed(lm, am, sym, blank, lm_only_scale=0.1, am_only_scale=0.2, boundary=bnd, rnnt_type=rnnt_type)
        l = ft.rnnt_loss_simple(lm, am, sym, blank, boundary=bnd, rnnt_type=rnnt_type, delay_penalty=0.3, reduction="none")
        return [x.cpu().numpy() for x in (*a, *b, l)]

    fused = run()
    monkeypatch.setenv("FTR_BUILDER_GEMM", "library")
    library = run()
    for f, l in zip(fused, library):
        assert f.shape == l.shape
        assert np.array_equal(np.isneginf(f), np.isneginf(l))
        fin = np.isfinite(l)
        if fin.any():      # (S > T with the modified type has no path: every loss is -inf)
            assert np.abs(f[fin] - l[fin]).max() <= 2e-5 * max(1.0, np.abs(l[fin]).max())


@pytest.mark.parametrize("rnnt_type", ["regular", "modified"])
@pytest.mark.parametrize("cfg", [(2, 72, 33, 12), (2, 129, 100, 20), (1, 200, 140, 260), (2, 68, 200, 8), (3, 64, 15, 36), (2, 100, 7, 600), (1, 76, 330, 16),
                                 (2, 132, 61, 1024), (2, 4, 3, 4), (1, 60, 0, 8), (1, 12, 700, 520), (1, 8, 4300, 12)])
def test_fused_d_am_kernel_matches_library_gemm_route(ft, dev, rnnt_type, cfg, monkeypatch):
    """The fused backward towards am (the default: W^T lm_probs as MFMA inside the kernel, scatter by symbol as a one-hot
    MFMA contraction) against the library route (FTR_BUILDER_BWD=library: library GEMM + epilogue kernel): gradients of the
    simple and of the smoothed loss w.r.t. am and lm, with boundaries and a non-uniform upstream gradient.  The scatter pass
    walks a per-workgroup list of the rows whose symbol lies in its 256 columns: the cases cover one and several column groups,
    lists built in several 256-row batches (S = 330, 700) and more listed rows than the list holds (S = 4300 in one column
    group: every row is walked, as before the list)."""
    B, T, S, C = cfg
    d = synthetic(23 + S, B, T, S, C, ragged=True)
    sym, bnd = _t(d["symbols"], dev), _t(d["boundary"], dev)
    blank = d["termination_symbol"]
    wts = torch.linspace(0.5, 1.5, B, device=dev)

    def grads():
        out = []
        for fn in (lambda lm, am: ft.rnnt_loss_simple(lm, am, sym, blank, boundary=bnd, rnnt_type=rnnt_type, reduction="none"),
                   lambda lm, am: ft.rnnt_loss_smoothed(lm, am, sym, blank, lm_only_scale=0.1, am_only_scale=0.2, boundary=bnd,
                                                        rnnt_type=rnnt_type, reduction="none")):
            lm = _t(d["lm"], dev).requires_grad_(True); am = _t(d["am"], dev).requires_grad_(True)
            loss = fn(lm, am)
            fin = torch.isfinite(loss)
            (loss[fin] * wts[fin]).sum().backward()
            out += [am.grad.cpu().numpy(), lm.grad.cpu().numpy()]
        return out

    monkeypatch.setenv("FTR_BUILDER_BWD", "library")
    library = grads()
    monkeypatch.setenv("FTR_BUILDER_BWD", "fused")
    fused = grads()
    for f, l in zip(fused, library):
        assert np.isfinite(f).all()
        assert np.abs(f - l).max() <= 2e-5 * max(1.0, np.abs(l).max())


@pytest.mark.parametrize("rows,C", [(1, 8), (63, 12), (64, 500), (65, 7), (1000, 33), (9632, 1024)])
def test_batch_statistics_kernels(ft, dev, rows, C):
    """ftr_colsum_weighted_f32 (two deterministic stages), ftr_rowdot_f32 and ftr_rowmax_exp_dot_f32 against float64 numpy:
    slabs that do not divide the rows, vector and scalar column paths."""
    from tf_fast_rnnt import _lib
    rng = np.random.default_rng(rows + C)
    x = rng.standard_normal((rows, C)).astype(np.float32)
    w = rng.standard_normal(rows).astype(np.float32)
    v = rng.random(C).astype(np.float32)
    tx, tw, tv = _t(x, dev), _t(w, dev), _t(v, dev)
    st = torch.cuda.current_stream().cuda_stream
    n = _lib.lib().ftr_colsum_weighted_workspace_floats(rows, C)
    ws = torch.empty(max(n, 1), device=dev); out = torch.empty(C, device=dev)
    for _ in range(2):       # twice: bit-reproducible
        _lib.call("ftr_colsum_weighted_f32", tx.data_ptr(), tw.data_ptr(), out.data_ptr(), ws.data_ptr(), n, rows, C, st)
        got = out.cpu().numpy()
        if _: assert np.array_equal(got, first)
        first = got
    want = (x.astype(np.float64) * w.astype(np.float64)[:, None]).sum(0)
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()) * max(1.0, np.sqrt(rows) / 8)
    dot = torch.empty(rows, device=dev)
    _lib.call("ftr_rowdot_f32", tx.data_ptr(), tv.data_ptr(), dot.data_ptr(), rows, C, st)
    np.testing.assert_allclose(dot.cpu().numpy(), x.astype(np.float64) @ v.astype(np.float64), rtol=1e-4, atol=1e-4)
    probs = torch.empty_like(tx); mx = torch.empty(rows, device=dev)
    _lib.call("ftr_rowmax_exp_dot_f32", tx.data_ptr(), probs.data_ptr(), mx.data_ptr(), tv.data_ptr(), dot.data_ptr(), rows, C, st)
    m = x.max(1); p = np.exp(x.astype(np.float64) - m[:, None])
    np.testing.assert_allclose(mx.cpu().numpy(), m)
    np.testing.assert_allclose(probs.cpu().numpy(), p, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(dot.cpu().numpy(), p @ v.astype(np.float64), rtol=1e-5)


def test_pruned_loss_routes_fuzz(ft, dev):
    """scripts/band_fuzz.py: 120 random pruned-loss problems (T from 1, one-row bands, r up to 20, both types, ragged
    boundaries): the band-native route against the full-lattice route.  (Found the modified-type end cell that sits one row
    above the last frame's band.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("band_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "band_fuzz.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.main(120, 20261004)


def test_fused_builder_kernels_fuzz(ft, dev):
    """scripts/builder_fuzz.py: 80 random builder problems (S from 0, T from 1, C = 4..316, both types, boundaries, penalty):
    the fused forward and the opt-in fused d am kernel against the library-GEMM route -- log-probs of the simple and smoothed
    builders and the gradients of both losses, 1e-4 normwise (the contractions sum in different orders)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("builder_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "builder_fuzz.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.main(80, 20261004)


def test_prune_ranges_fuzz_bit_exact(ft, dev, oracle):
    """scripts/prune_fuzz.py: 120 random inputs (ties, ragged boundaries, window lengths 1 .. S+3 incl. the > 16 generic
    kernel, both types): ranges bit for bit against the oracle."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("prune_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "prune_fuzz.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.main(120, 20261004)


def test_whole_pipeline_fuzz_against_oracle(ft, dev, oracle):
    """40 random small problems through the whole package path (simple loss with occupancies -> ranges -> gather -> pruned
    loss + gradient) against the float64 oracle fed with the SAME ranges: loss 1e-4, gradient 2e-4 normwise; regular and
    modified, ragged boundaries, r from 1."""
    rng = np.random.default_rng(77)
    for it in range(40):
        B = int(rng.integers(1, 4)); T = int(rng.choice([3, 8, 17, 40])); S = int(rng.choice([1, 2, 5, 12])); C = int(rng.choice([4, 5, 8, 10]))
        r = int(rng.choice([1, 2, 3, 5, 8])); rt = "modified" if rng.integers(0, 2) else "regular"
        if rt == "modified" and S > T: S = T
        d = synthetic(int(rng.integers(1, 10**6)), B, T, S, C, ragged=bool(rng.integers(0, 2)))
        blank = d["termination_symbol"]
        am, lm, sym, bd = (_t(d[k], dev) for k in ("am", "lm", "symbols", "boundary"))
        _, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, blank, bd, rt, reduction="sum", calc_gradients=True)
        ranges = ft.get_rnnt_prune_ranges(gx, gy, bd, r)
        am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
        logits = torch.tanh(am_p + lm_p).detach().requires_grad_(True)
        loss = ft.rnnt_loss_pruned(logits, sym, ranges, blank, bd, rt, 0.05, "none")
        fin = torch.isfinite(loss)
        if fin.any(): loss[fin].sum().backward()
        o_loss, o_g = oracle.rnnt_loss_pruned_grad(logits.detach().cpu().numpy(), d["symbols"], ranges.cpu().numpy(), blank, d["boundary"],
                                                   rnnt_type=rt, delay_penalty=0.05, reduction="none", dtype=np.float64)
        f = fin.cpu().numpy()
        assert np.array_equal(f, np.isfinite(o_loss)), (it, B, T, S, C, r, rt)
        if f.any():
            np.testing.assert_allclose(loss.detach().cpu().numpy()[f], o_loss[f], rtol=1e-4, atol=1e-5)
            assert max_rel(logits.grad.cpu().numpy()[f], o_g[f]) <= 2e-4, (it, B, T, S, C, r, rt)


def test_gemm_kernel_selection(ft, dev, monkeypatch):
    """The library GEMMs of the builders' backward (ftr_normalizer_gemm_f32, rocBLAS): against torch.bmm for all three kinds;
    the kernel is measured at the second call with a shape (default), which changes which library kernel runs, not the
    result (1e-5 normwise); FTR_GEMM_TUNE=off leaves a new shape unmeasured; a recorded choice can be carried over."""
    g = torch.Generator(device="cpu").manual_seed(3)
    B, T, S, C = 3, 72, 33, 40
    lm_probs = torch.rand((B, S + 1, C), generator=g).to(dev); am_probs = torch.rand((B, T, C), generator=g).to(dev)
    W = torch.randn((B, S + 1, T), generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    from tf_fast_rnnt.rnnt_loss import _gemm
    want = (torch.bmm(lm_probs, am_probs.transpose(1, 2)), torch.bmm(W, am_probs), torch.bmm(W.transpose(1, 2), lm_probs))
    monkeypatch.setenv("FTR_GEMM_TUNE", "second")
    for kind, (x, y) in enumerate(((lm_probs, am_probs), (W, am_probs), (W, lm_probs))):
        first = _gemm(kind, x, y, B, T, S, C, st)
        ch = ft.normalizer_gemm_choice(kind, B, T, S, C)
        assert ch is not None and ch["candidates"] == -1 and ch["solution"] == 0       # seen once: not measured yet
        second = _gemm(kind, x, y, B, T, S, C, st)                                     # measures, then runs the winner
        ch = ft.normalizer_gemm_choice(kind, B, T, S, C)
        assert ch["candidates"] >= 0 and ch["us_default"] > 0
        if ch["solution"]:
            assert ch["us"] <= 0.97 * ch["us_default"]
        for got in (first, second, _gemm(kind, x, y, B, T, S, C, st)):
            assert got.shape == want[kind].shape
            assert (got - want[kind]).abs().max().item() <= 1e-5 * max(1.0, want[kind].abs().max().item())
    # off: a new shape stays with the library's own kernel however often it runs
    monkeypatch.setenv("FTR_GEMM_TUNE", "off")
    for _ in range(3):
        _gemm(1, W[:2], am_probs[:2], 2, T, S, C, st)
    assert ft.normalizer_gemm_choice(1, 2, T, S, C)["candidates"] == -1
    # a recorded choice for a shape that has not run (an index the library rejects falls back to its own kernel)
    ft.set_normalizer_gemm_choice(1, 1, T, S, C, ft.normalizer_gemm_choice(1, B, T, S, C)["solution"])
    ft.set_normalizer_gemm_choice(2, 1, T, S, C, 2147483)
    for kind, y in ((1, am_probs), (2, lm_probs)):
        got = _gemm(kind, W[:1], y[:1], 1, T, S, C, st)
        assert (got - want[kind][:1]).abs().max().item() <= 1e-5 * max(1.0, want[kind].abs().max().item())
    assert ft.normalizer_gemm_choice(2, 1, T, S, C)["solution"] == 0
    # the switch function only moves the environment variable
    ft.tune_normalizer_gemms(False)
    assert os.environ["FTR_GEMM_TUNE"] == "off"
    ft.tune_normalizer_gemms(True, when="first")
    assert os.environ["FTR_GEMM_TUNE"] == "first"
    monkeypatch.delenv("FTR_GEMM_TUNE")


@pytest.mark.parametrize("shape", [(3, 40, 12, 20), (2, 65, 0, 7), (1, 1, 5, 516)])
def test_rowmax_exp_pair_is_the_two_single_launches(ft, dev, shape):
    """ftr_rowmax_exp_pair_f32 (am and lm of the simple builder in one launch) against two ftr_rowmax_exp_f32 calls: same bits."""
    B, T, S, C = shape
    g = torch.Generator(device="cpu").manual_seed(C)
    am = (3.0 * torch.randn((B * T, C), generator=g)).to(dev); lm = (3.0 * torch.randn((B * (S + 1), C), generator=g)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    L = ft._lib
    ap, lp = torch.empty_like(am), torch.empty_like(lm)
    amx, lmx = torch.empty(B * T, device=dev), torch.empty(B * (S + 1), device=dev)
    L.call("ftr_rowmax_exp_f32", am.data_ptr(), ap.data_ptr(), amx.data_ptr(), B * T, C, st)
    L.call("ftr_rowmax_exp_f32", lm.data_ptr(), lp.data_ptr(), lmx.data_ptr(), B * (S + 1), C, st)
    ap2, lp2 = torch.full_like(am, float("nan")), torch.full_like(lm, float("nan"))
    amx2, lmx2 = torch.full_like(amx, float("nan")), torch.full_like(lmx, float("nan"))
    L.call("ftr_rowmax_exp_pair_f32", am.data_ptr(), ap2.data_ptr(), amx2.data_ptr(), B * T, lm.data_ptr(), lp2.data_ptr(), lmx2.data_ptr(),
           B * (S + 1), C, st)
    assert torch.equal(ap, ap2) and torch.equal(lp, lp2) and torch.equal(amx, amx2) and torch.equal(lmx, lmx2)
