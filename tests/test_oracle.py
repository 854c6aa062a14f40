"""CPU tests of the oracle itself (no GPU): pinned against the reference's docstring vectors where they
exist, cross-checked independently (brute force over paths, float64 torch autograd DP, invariants)
everywhere else, and frozen by the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

from helpers import random_lattice, reference_test_recipe, synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- known-answer vectors held by the reference itself ---------------------------------------------
def test_monotonic_lower_bound_docstring_vectors(oracle):
    # rnnt_loss.py:561-563
    assert np.array_equal(oracle.monotonic_lower_bound(np.array([0, 2, 1, 3, 6, 5, 8])), [0, 1, 1, 3, 5, 5, 8])
    # rnnt_loss.py:566-574
    x = np.array([[12, 18, 5, 4, 18, 17], [11, 14, 14, 3, 10, 4], [19, 3, 8, 13, 7, 19]])
    want = np.array([[4, 4, 4, 4, 17, 17], [3, 3, 3, 3, 4, 4], [3, 3, 7, 7, 7, 19]])
    assert np.array_equal(oracle.monotonic_lower_bound(x), want)


def test_roll_by_shifts_docstring_vector(oracle):
    # rnnt_loss.py:823-834
    src = np.arange(15).reshape(1, 3, 5)
    want = np.array([[[4, 0, 1, 2, 3], [8, 9, 5, 6, 7], [12, 13, 14, 10, 11]]])
    assert np.array_equal(oracle.roll_by_shifts(src, np.array([[1, 2, 3]])), want)


def test_cummin_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    x = rng.integers(-50, 50, (7, 133)).astype(np.int32)
    assert np.array_equal(oracle.cummin(x), np.minimum.accumulate(x, axis=1))


def test_ranges_docstring_properties(oracle):
    """rnnt_loss.py:663-677: ranges[:,0] monotone from 0 to len(symbols)-s_range, steps < s_range."""
    d = reference_test_recipe(1234, 2, 10, 7, 4)
    _, (gx, gy) = oracle.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], d["termination_symbol"], d["boundary"],
                                          reduction="none", calc_gradients=True)
    for r in (2, 3, 5):
        ranges = oracle.get_rnnt_prune_ranges(gx, gy, d["boundary"], r)
        assert np.array_equal(ranges, oracle.get_rnnt_prune_ranges_numpy(gx, gy, d["boundary"], r))
        s0 = ranges[:, :, 0]
        assert (s0[:, 0] == 0).all()
        assert (np.diff(s0, axis=1) >= 0).all() and (np.diff(s0, axis=1) < r).all()
        for b in range(2):
            te, se = d["boundary"][b, 3], d["boundary"][b, 2]
            assert s0[b, te - 1] == max(se - r + 1, 0)


# ---- LogAdd / safe_exp edge semantics ----------------------------------------------------------------
def test_logadd_edges(oracle):
    L = oracle.lib()
    inf = float("inf")
    assert L.oracle_logadd_f32(-inf, -inf) == -inf                    # "return the larger one" branch
    assert L.oracle_logadd_f32(-inf, 1.5) == 1.5 and L.oracle_logadd_f32(1.5, -inf) == 1.5
    np.testing.assert_allclose(L.oracle_logadd_f32(0.0, 0.0), np.log(2.0), rtol=1e-6)
    assert L.oracle_safe_exp_f32(-inf) == 0.0 and L.oracle_safe_exp_f32(float("nan")) == 0.0
    assert L.oracle_safe_exp_f32(1000.0) == 0.0                        # overflow -> 0, not inf


# ---- independent checks of the recursion ---------------------------------------------------------------
@pytest.mark.parametrize("modified", [False, True])
def test_forward_vs_brute_force(oracle, modified):
    px, py, bd = random_lattice(0, 4, 3, 4, modified=modified, ragged=True, begin_offsets=True)
    bd[0] = [0, 0, 3, 4]
    ans, _ = oracle.mi_forward(px, py, bd)
    want = oracle.brute_force_mi(px, py, bd, modified=modified)
    np.testing.assert_allclose(ans, want, rtol=1e-5, atol=1e-6)
    ans64, _ = oracle.mi_forward(px, py, bd, dtype=np.float64)
    np.testing.assert_allclose(ans64, want, rtol=1e-12, atol=1e-12)


def _torch_dp(px, py, bd, modified):
    """Independent float64 DP with torch.logaddexp; gradients by autograd."""
    px = torch.tensor(px, dtype=torch.float64, requires_grad=True)
    py = torch.tensor(py, dtype=torch.float64, requires_grad=True)
    B, S, _ = px.shape
    T = py.shape[2]
    tot = []
    for b in range(B):
        sb, tb, se, te = [int(v) for v in bd[b]]
        p = {}
        for s in range(sb, se + 1):
            for t in range(tb, te + 1):
                if s == sb and t == tb:
                    p[s, t] = torch.zeros((), dtype=torch.float64)
                    continue
                terms = []
                if modified:
                    if s > sb and t > tb and p[s - 1, t - 1] is not None:
                        terms.append(p[s - 1, t - 1] + px[b, s - 1, t - 1])
                elif s > sb and p[s - 1, t] is not None:
                    terms.append(p[s - 1, t] + px[b, s - 1, t])
                if t > tb and p[s, t - 1] is not None:
                    terms.append(p[s, t - 1] + py[b, s, t - 1])
                p[s, t] = torch.logsumexp(torch.stack(terms), 0) if terms else None   # None = unreachable
        tot.append(p[se, te])
    ans = torch.stack(tot)
    ans.sum().backward()
    return ans.detach().numpy(), px.grad.numpy(), py.grad.numpy()


@pytest.mark.parametrize("modified", [False, True])
def test_backward_vs_autograd(oracle, modified):
    px, py, bd = random_lattice(1, 3, 5, 7, modified=modified, ragged=True)
    bd[1] = [1, 2, 4, 6]
    ans, p = oracle.mi_forward(px, py, bd)
    gx, gy, chk = oracle.mi_backward(px, py, bd, p)
    a, ax, ay = _torch_dp(px, py, bd, modified)
    np.testing.assert_allclose(ans, a, rtol=1e-5)
    np.testing.assert_allclose(gx, ax, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(gy, ay, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(chk, 1.0, rtol=1e-4)     # the reference's ans_grad self-check


def test_occupancy_invariants(oracle):
    d = synthetic(3, 2, 60, 20, 30, ragged=True)
    _, (gx, gy) = oracle.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], d["termination_symbol"], d["boundary"],
                                          reduction="none", calc_gradients=True)
    for b in range(2):
        se, te = d["boundary"][b, 2], d["boundary"][b, 3]
        np.testing.assert_allclose(gy[b, :se + 1, :te].sum(axis=0), 1.0, rtol=5e-4)   # one blank per frame
        np.testing.assert_allclose(gx[b, :se, :te + 1].sum(axis=1), 1.0, rtol=5e-4)   # every symbol once
        assert not gx[b, se:].any() and not gy[b, :, te:].any()


def test_logprob_builders_are_normalised(oracle):
    """exp(px) + exp(py) summed appropriately: for the simple builder, the symbol and blank probabilities of a
    cell come from one softmax over C, so exp(py[s,t]) + sum over all symbols would be 1; check the blank and
    the chosen-symbol entries against a direct float64 softmax."""
    d = synthetic(4, 2, 9, 5, 11)
    px, py = oracle.get_rnnt_logprobs(d["lm"], d["am"], d["symbols"], d["termination_symbol"])
    joint = d["am"][:, None, :, :].astype(np.float64) + d["lm"][:, :, None, :].astype(np.float64)   # [B,S+1,T,C]
    logsm = joint - np.log(np.exp(joint).sum(-1, keepdims=True))
    np.testing.assert_allclose(py, logsm[..., d["termination_symbol"]], rtol=1e-5, atol=1e-5)
    for b in range(2):
        for s in range(5):
            np.testing.assert_allclose(px[b, s, :9], logsm[b, s, :, d["symbols"][b, s]], rtol=1e-5, atol=1e-5)
    assert np.isneginf(px[:, :, 9]).all()


def test_pruned_band_equals_opwise(oracle):
    """The C band arithmetic (used for the logits gradient) against the op-by-op numpy restatement."""
    d = reference_test_recipe(1234, 2, 30, 9, 8)
    _, (gx, gy) = oracle.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], d["termination_symbol"], d["boundary"],
                                          reduction="none", calc_gradients=True)
    r = 3
    ranges = oracle.get_rnnt_prune_ranges(gx, gy, d["boundary"], r)
    am_p, lm_p = oracle.do_rnnt_pruning(d["am"], d["lm"], ranges)
    logits = (am_p + lm_p).astype(np.float32)
    px, py = oracle.get_rnnt_logprobs_pruned(logits, d["symbols"], ranges, d["termination_symbol"], None)
    lse, pxb, pyb = oracle.pruned_band_fwd(logits, d["symbols"], ranges, d["termination_symbol"])
    B, T, _ = ranges.shape
    S = d["S"]
    for b in range(B):
        for t in range(T):
            for k in range(r):
                s = ranges[b, t, k]
                np.testing.assert_allclose(py[b, s, t], pyb[b, t, k], rtol=1e-6, atol=1e-6)
                if s < S:
                    np.testing.assert_allclose(px[b, s, t], pxb[b, t, k], rtol=1e-6, atol=1e-6)
    # gradient of the pruned loss w.r.t. logits against float64 autograd through an independent DP
    loss, g = oracle.rnnt_loss_pruned_grad(logits, d["symbols"], ranges, d["termination_symbol"], d["boundary"], reduction="sum")
    lg = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    logp = lg - torch.logsumexp(lg, dim=3, keepdim=True)
    total = 0.0
    for b in range(B):
        se, te = int(d["boundary"][b, 2]), int(d["boundary"][b, 3])
        neg = torch.tensor(-float("inf"), dtype=torch.float64)
        p = {}
        for s in range(se + 1):
            for t in range(te + 1):
                if s == 0 and t == 0:
                    p[s, t] = torch.zeros((), dtype=torch.float64); continue
                terms = []
                if s > 0 and t < te:
                    k = s - 1 - int(ranges[b, t, 0])
                    if 0 <= k < r:
                        terms.append(p[s - 1, t] + logp[b, t, k, int(d["symbols"][b, s - 1])])
                if t > 0:
                    k = s - int(ranges[b, t - 1, 0])
                    if 0 <= k < r:
                        terms.append(p[s, t - 1] + logp[b, t - 1, k, d["termination_symbol"]])
                p[s, t] = torch.logsumexp(torch.stack(terms), 0) if terms else neg
        total = total - p[se, te]
    total.backward()
    np.testing.assert_allclose(loss, total.item(), rtol=1e-5)
    np.testing.assert_allclose(g, lg.grad.numpy(), rtol=2e-3, atol=2e-6)


# ---- golden fixtures (generated from the oracle by tests/golden/make_golden.py) --------------------------
@pytest.mark.parametrize("name", ["c1_B2_T8_S4_C16", "seed1234_B2_T10_S7_C4", "seed12345_B2_T200_S50_C50"])
def test_oracle_reproduces_golden(oracle, name):
    from golden.make_golden import compute
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = compute({k: z[k] for k in ("am", "lm", "symbols", "boundary")}, int(z["termination_symbol"]),
                  [int(v) for v in z["s_ranges"]])
    for k, v in out.items():
        if v.dtype.kind in "iu":
            assert np.array_equal(v, z[k]), k
        else:
            np.testing.assert_allclose(v, z[k], rtol=2e-5, atol=1e-6, err_msg=k)
