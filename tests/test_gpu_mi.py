"""Parity of the native mutual-information ops against the oracle, through the C ABI (via the ctypes
binding the package uses).  Tolerances (north_star: "loss and px/py gradients within 1e-4 relative"):

* ans / loss: elementwise rtol 1e-4 against the float32 oracle (observed ~1e-6).
* px_grad / py_grad: normwise relative error (max|diff| / max|ref|) <= 1e-4 against the float32 oracle
  on lattices where float32 log-domain arithmetic itself is accurate to that level, and against the
  float64 oracle everywhere.  On long lattices (|p| of several thousand) the reference arithmetic is
  itself only ~1e-3 accurate (float32 oracle vs float64 oracle: 7e-4 normwise at T=1000,S=200,C=500,
  measured in DESIGN.md), so there the bound versus the float32 oracle is the oracle's own error.
"""
import numpy as np
import pytest
import torch

from helpers import assert_parity, max_rel, random_lattice

pytestmark = pytest.mark.gpu

def _run(ft, dev, px, py, bd, impl, need_grads=True):
    """impl "wavefront": the product kernels through the package; "plain": the reference's arithmetic on the device, from the
    test-only diag library (tests/diag.py)."""
    from tf_fast_rnnt.mutual_information import mi_forward_backward
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if impl == "plain":
        import diag
        ans, gx, gy, chk = diag.plain_forward_backward(t(px), t(py), t(bd), need_grads)
    else:
        ans, gx, gy, chk = mi_forward_backward(t(px), t(py), t(bd), need_grads, return_ans_grad_check=True)
    torch.cuda.synchronize()
    return [ans.cpu().numpy()] + [None if g is None else g.cpu().numpy() for g in (gx, gy, chk)]


def test_selftest(ft, dev):
    scratch = torch.zeros(4096, dtype=torch.int32, device=dev)
    ft._lib.check(ft._lib.lib().ftr_selftest(scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "selftest")
    torch.cuda.synchronize()
    assert int(scratch[0].item()) == 1


@pytest.mark.parametrize("impl", ["wavefront", "plain"])
@pytest.mark.parametrize("modified", [False, True])
@pytest.mark.parametrize("shape", [(2, 0, 5), (2, 5, 1), (1, 0, 1), (2, 1, 40), (2, 4, 8), (3, 1, 1), (3, 7, 10), (4, 50, 200), (2, 63, 70), (2, 64, 65),
                                   (2, 65, 33), (3, 130, 90), (2, 200, 257), (1, 300, 40), (2, 383, 150), (2, 400, 130), (1, 1100, 70)])
def test_mi_parity_f32_oracle(ft, dev, oracle, impl, modified, shape):
    B, S, T = shape
    if impl == "plain" and S + 1 > 1024:
        pytest.skip("the plain family (one thread per row, one workgroup) does not cover this many rows")
    px, py, bd = random_lattice(100 + S + T, B, S, T, modified=modified, ragged=True)
    ans, gx, gy, chk = _run(ft, dev, px, py, bd, impl)
    o_ans, o_p = oracle.mi_forward(px, py, bd)
    o_gx, o_gy, o_chk = oracle.mi_backward(px, py, bd, o_p)
    np.testing.assert_allclose(ans, o_ans, rtol=1e-4, atol=1e-5)
    # utterances with no valid path (ans = -inf; e.g. modified with s_end > t_end) have no defined gradient:
    # the reference propagates its seed through unreachable cells with term = exp(0) there.  Compare the rest.
    ok = np.isfinite(o_ans)
    assert not np.isnan(gx).any() and not np.isnan(gy).any()
    if max(max_rel(gx[ok], o_gx[ok]), max_rel(gy[ok], o_gy[ok])) > 1e-4:
        # long lattice: float32 log-domain noise regime, see the module docstring -> float64 leg of the rule
        a64, p64 = oracle.mi_forward(px, py, bd, dtype=np.float64)
        gx64, gy64, _ = oracle.mi_backward(px, py, bd, p64, dtype=np.float64)
        assert_parity(gx[ok], o_gx[ok], gx64[ok].astype(np.float32), what="px_grad")
        assert_parity(gy[ok], o_gy[ok], gy64[ok].astype(np.float32), what="py_grad")
    # the reference's self-check (mutual_information_cuda.cu:510-514): recomputed ans_grad == seed
    nonempty = (bd[:, 2] >= bd[:, 0]) & (bd[:, 3] >= bd[:, 1]) & ok
    np.testing.assert_allclose(chk[nonempty], 1.0, rtol=2e-4)
    # zeros outside the boundary rectangle, exactly
    for b in range(B):
        se, te = bd[b, 2], bd[b, 3]
        assert not gx[b, se:, :].any() and not gy[b, se + 1:, :].any()
        assert not gx[b, :, te + 1:].any() and not gy[b, :, te:].any()


@pytest.mark.parametrize("impl", ["wavefront", "plain"])
@pytest.mark.parametrize("modified", [False, True])
def test_mi_begin_offsets_and_empty(ft, dev, oracle, impl, modified):
    B, S, T = 5, 20, 37
    px, py, bd = random_lattice(5, B, S, T, modified=modified, ragged=True, begin_offsets=True)
    bd[1] = [3, 5, 3, 5]          # single cell: ans = 0
    bd[2] = [0, 0, 0, T]          # no symbols at all
    ans, gx, gy, chk = _run(ft, dev, px, py, bd, impl)
    o_ans, o_p = oracle.mi_forward(px, py, bd)
    o_gx, o_gy, _ = oracle.mi_backward(px, py, bd, o_p)
    np.testing.assert_allclose(ans, o_ans, rtol=1e-4, atol=1e-5)
    assert ans[1] == 0.0
    ok = np.isfinite(o_ans)
    assert max_rel(gx[ok], o_gx[ok]) <= 1e-4 and max_rel(gy[ok], o_gy[ok]) <= 1e-4


@pytest.mark.parametrize("impl", ["wavefront", "plain"])
@pytest.mark.parametrize("modified", [False, True])
@pytest.mark.parametrize("frac", [0.02, 0.3])
def test_mi_neg_inf_entries(ft, dev, oracle, impl, modified, frac):
    """-inf inside px/py (the pruned lattices are >97% -inf): same finite/-inf pattern of ans, no NaN."""
    B, S, T = 6, 30, 45
    px, py, bd = random_lattice(11, B, S, T, modified=modified, neg_inf_frac=frac, ragged=False)
    ans, gx, gy, _ = _run(ft, dev, px, py, bd, impl)
    o_ans, o_p = oracle.mi_forward(px, py, bd)
    o_gx, o_gy, _ = oracle.mi_backward(px, py, bd, o_p)
    assert np.array_equal(np.isfinite(ans), np.isfinite(o_ans))
    fin = np.isfinite(o_ans)
    np.testing.assert_allclose(ans[fin], o_ans[fin], rtol=1e-4, atol=1e-5)
    assert not np.isnan(gx).any() and not np.isnan(gy).any()
    if fin.any():   # utterances with no surviving path have reference-defined garbage gradients
        assert max_rel(gx[fin], o_gx[fin]) <= 1e-4 and max_rel(gy[fin], o_gy[fin]) <= 1e-4


def test_mi_boundary_none(ft, dev, oracle):
    px, py, _ = random_lattice(3, 2, 9, 14, ragged=False)
    ans, gx, gy, _ = _run(ft, dev, px, py, None, "wavefront")
    o_ans, o_p = oracle.mi_forward(px, py, None)
    o_gx, o_gy, _ = oracle.mi_backward(px, py, None, o_p)
    np.testing.assert_allclose(ans, o_ans, rtol=1e-4)
    assert max_rel(gx, o_gx) <= 1e-4 and max_rel(gy, o_gy) <= 1e-4


@pytest.mark.parametrize("impl", ["wavefront", "plain"])
def test_mi_vs_float64_oracle_long(ft, dev, oracle, impl):
    """Long lattice with realistic magnitudes (px,py ~ log(1/C)): compare with the float64 oracle and
    require the native result to be no worse than the float32 reference arithmetic."""
    B, S, T = 2, 150, 700
    rng = np.random.default_rng(0)
    px = (rng.standard_normal((B, S, T + 1)) - 6.0).astype(np.float32)
    py = (rng.standard_normal((B, S + 1, T)) - 6.0).astype(np.float32)
    bd = np.array([[0, 0, S, T], [0, 0, S - 11, T - 50]], dtype=np.int32)
    ans, gx, gy, _ = _run(ft, dev, px, py, bd, impl)
    a64, p64 = oracle.mi_forward(px, py, bd, dtype=np.float64)
    gx64, gy64, _ = oracle.mi_backward(px, py, bd, p64, dtype=np.float64)
    a32, p32 = oracle.mi_forward(px, py, bd)
    gx32, gy32, _ = oracle.mi_backward(px, py, bd, p32)
    np.testing.assert_allclose(ans, a64, rtol=1e-4)
    ref_err = max(max_rel(gx32, gx64), max_rel(gy32, gy64))
    err = max(max_rel(gx, gx64), max_rel(gy, gy64))
    assert err <= max(1e-4, 1.5 * ref_err), (err, ref_err)


@pytest.mark.parametrize("modified", [False, True])
def test_operand_shift_is_invisible(ft, dev, oracle, modified):
    """The kernels subtract per-utterance constants from px / py before the recursion (csrc/ftr_common.h, Shift) and add
    what that took out of `ans` back.  Seen from outside nothing may depend on it: adding constants (cx to every px, cy
    to every py) to the INPUTS moves ans by (px steps) cx + (py steps) cy of a complete path and leaves the occupancies
    alone -- also for utterances with begin offsets, -inf entries, huge finite stand-ins for -inf (excluded from the
    sampled means) and one lattice whose px are all -inf."""
    B, S, T = 4, 90, 300
    px, py, bd = random_lattice(77, B, S, T, modified=modified, neg_inf_frac=0.02, ragged=True, begin_offsets=True)
    px[1][px[1] < -2.5] = -1.0e20                 # stand-ins for -inf
    px[3] = -np.inf                               # no px at all: only the utterance with s_end == s_begin has a path
    bd[3] = (5, 3, 5, T)
    py[3, 5, :] = np.float32(-1.0) - np.abs(py[0, 5, :]).clip(0, 5)   # ... along finite blanks
    ans0, gx0, gy0, _ = _run(ft, dev, px, py, bd, "wavefront")
    o_ans, (o_gx, o_gy) = oracle.mutual_information_recursion(px, py, bd, True, np.float64)
    ok = np.isfinite(o_ans)
    assert ok[3] and np.array_equal(np.isfinite(ans0), ok)
    np.testing.assert_allclose(ans0[ok], o_ans[ok], rtol=3e-6, atol=3e-5)
    assert max_rel(gx0[ok], o_gx[ok]) <= 2e-5 and max_rel(gy0[ok], o_gy[ok]) <= 2e-5
    for cx, cy in ((-37.5, 11.25), (300.0, -400.0)):
        ans1, gx1, gy1, _ = _run(ft, dev, (px + np.float32(cx)).astype(np.float32), (py + np.float32(cy)).astype(np.float32), bd, "wavefront")
        nx = (bd[:, 2] - bd[:, 0]).astype(np.float64)
        ny = (bd[:, 3] - bd[:, 1]).astype(np.float64) - (nx if modified else 0.0)
        np.testing.assert_allclose(ans1[ok], (o_ans + nx * cx + ny * cy)[ok], rtol=3e-6, atol=3e-5)
        assert max_rel(gx1[ok], o_gx[ok]) <= 1e-4 and max_rel(gy1[ok], o_gy[ok]) <= 1e-4   # px + cx rounds the inputs themselves


@pytest.mark.parametrize("kind", ["sharp", "blank_heavy", "tilted"])
def test_accuracy_on_structured_lattices(ft, dev, oracle, kind):
    """Accuracy against the float64 oracle where the sampled means say little about the paths that carry the
    occupancy: a sharp (trained-looking) model whose alignment has px ~ -0.1 / py ~ -0.05 on it and -10 / -3 off it; a
    blank-heavy model (py ~ -0.1, px ~ -8 everywhere); and independent cells with very different px and py means."""
    B, S, T = 2, 200, 1000
    rng = np.random.default_rng(5)
    if kind == "sharp":
        px = (rng.standard_normal((B, S, T + 1)) - 10.0).astype(np.float32)
        py = (rng.standard_normal((B, S + 1, T)) - 3.0).astype(np.float32)
        for b in range(B):
            ts = np.sort(np.clip(np.round((np.arange(S) + 0.5) * T / S).astype(int) + rng.integers(-3, 4, S), 0, T - 1))
            prev = 0
            for s_ in range(S + 1):
                end = ts[s_] if s_ < S else T
                py[b, s_, prev:end] = -0.05
                if s_ < S:
                    px[b, s_, end] = -0.1
                prev = end
    elif kind == "blank_heavy":
        px = (0.5 * rng.standard_normal((B, S, T + 1)) - 8.0).astype(np.float32)
        py = (0.05 * rng.standard_normal((B, S + 1, T)) - 0.1).astype(np.float32)
    else:
        px = (rng.standard_normal((B, S, T + 1)) - 2.0).astype(np.float32)
        py = (rng.standard_normal((B, S + 1, T)) - 9.0).astype(np.float32)
    px[:, :, T] = -np.inf
    bd = np.zeros((B, 4), np.int32); bd[:, 2] = S; bd[:, 3] = T
    ans, gx, gy, _ = _run(ft, dev, px, py, bd, "wavefront")
    a64, (gx64, gy64) = oracle.mutual_information_recursion(px, py, bd, True, np.float64)
    # (the sharp lattice adds the SAME blank log-probability at every step of its alignment, so the roundings of its running
    # sum do not average out; measured 6e-6 relative, 1e-7 ... 5e-8 on the other two)
    np.testing.assert_allclose(ans, a64, rtol=2e-5 if kind == "sharp" else 2e-6)
    assert max_rel(gx, gx64) <= 2e-5 and max_rel(gy, gy64) <= 2e-5, (kind, max_rel(gx, gx64), max_rel(gy, gy64))


def test_mi_autograd(ft, dev, oracle):
    """The registered gradient (__init__.py:154-162): d(sum_b w_b ans_b)/d px = w_b * px_grad."""
    px, py, bd = random_lattice(21, 3, 6, 9, ragged=True)
    tpx = torch.from_numpy(px).to(dev).requires_grad_(True)
    tpy = torch.from_numpy(py).to(dev).requires_grad_(True)
    w = torch.tensor([0.5, -2.0, 3.0], device=dev)
    ans = ft.mutual_information_recursion(tpx, tpy, torch.from_numpy(bd).to(dev))
    (ans * w).sum().backward()
    o_ans, o_p = oracle.mi_forward(px, py, bd)
    o_gx, o_gy, _ = oracle.mi_backward(px, py, bd, o_p)
    wn = w.cpu().numpy().reshape(-1, 1, 1)
    assert max_rel(tpx.grad.cpu().numpy(), wn * o_gx) <= 1e-4
    assert max_rel(tpy.grad.cpu().numpy(), wn * o_gy) <= 1e-4


def test_mi_full_size_properties(ft, dev):
    """BASELINE config c3 lattice size (B=32,T=1000,S=200): size-independent properties of the occupancies
    (SURVEY.md section 7.1): sum_s py_grad[:,t] = 1 for t < t_end, sum_t px_grad[s,:] = 1 for s < s_end,
    recomputed ans_grad = 1, and agreement of the two kernel families."""
    B, S, T = 32, 200, 1000
    g = torch.Generator(device="cpu").manual_seed(0)
    px = (torch.randn((B, S, T + 1), generator=g) - 6.0)
    py = (torch.randn((B, S + 1, T), generator=g) - 6.0)
    bd = torch.zeros((B, 4), dtype=torch.int32)
    bd[:, 2] = S; bd[:, 3] = T
    bd[1, 2] = 77; bd[1, 3] = 513; bd[2, 2] = 199; bd[2, 3] = 999
    px = px.scatter(2, bd[:, 3].long().reshape(B, 1, 1).expand(B, S, 1), float("-inf"))
    outs = {}
    for impl in ("wavefront", "plain"):
        outs[impl] = _run(ft, dev, px.numpy(), py.numpy(), bd.numpy(), impl)
    for impl, (ans, gx, gy, chk) in outs.items():
        tol = 1e-2 if impl == "plain" else 1e-4     # plain = reference arithmetic: its normalisation drifts (3e-3 here)
        for b in range(B):
            se, te = int(bd[b, 2]), int(bd[b, 3])
            np.testing.assert_allclose(gy[b, :se + 1, :te].sum(axis=0), 1.0, rtol=tol)
            np.testing.assert_allclose(gx[b, :se, :te + 1].sum(axis=1), 1.0, rtol=tol)
        np.testing.assert_allclose(chk, 1.0, rtol=tol)
    np.testing.assert_allclose(outs["wavefront"][0], outs["plain"][0], rtol=1e-5)
    # plain = the reference arithmetic, whose float32 noise at this size is ~7e-3
    assert max_rel(outs["wavefront"][1], outs["plain"][1]) <= 2e-2
    assert max_rel(outs["wavefront"][2], outs["plain"][2]) <= 2e-2


def test_cummin(ft, dev, oracle):
    rng = np.random.default_rng(0)
    for rows, cols in [(1, 1), (3, 6), (5, 64), (7, 65), (32, 1000), (2, 4097)]:
        x = rng.integers(-1000, 1000, (rows, cols)).astype(np.int32)
        out = ft.cummin(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert np.array_equal(out, oracle.cummin(x))
        assert np.array_equal(out, np.minimum.accumulate(x, axis=1))


def test_mi_long_form_properties(ft, dev):
    """BASELINE config c5 lattice (B=8, S=1000, T=8000; 16 bands, 9000-step lattice): occupancy invariants of the default
    family at a size the CPU oracle is not asked to follow -- every frame is left exactly once (sum_s py_grad[:,t] = 1),
    every symbol is emitted exactly once (sum_t px_grad[s,:] = 1), the ans_grad self check returns the seed, zeros
    outside a ragged boundary; 1e-3 (float32 flows over 9000 steps; observed ~1e-5)."""
    B, S, T = 8, 1000, 8000
    g = torch.Generator(device="cpu").manual_seed(5)
    px = (torch.randn((B, S, T + 1), generator=g) - 6.0)
    py = (torch.randn((B, S + 1, T), generator=g) - 6.0)
    bd = torch.zeros((B, 4), dtype=torch.int32)
    bd[:, 2] = S; bd[:, 3] = T
    bd[1, 2] = 511; bd[1, 3] = 4097; bd[2, 2] = 64; bd[2, 3] = 7999
    px = px.scatter(2, bd[:, 3].long().reshape(B, 1, 1).expand(B, S, 1), float("-inf"))
    ans, gx, gy, chk = _run(ft, dev, px.numpy(), py.numpy(), bd.numpy(), "wavefront")
    assert np.isfinite(ans).all()
    np.testing.assert_allclose(chk, 1.0, rtol=1e-3)
    for b in range(B):
        se, te = int(bd[b, 2]), int(bd[b, 3])
        np.testing.assert_allclose(gy[b, :se + 1, :te].sum(axis=0, dtype=np.float64), 1.0, rtol=1e-3)
        np.testing.assert_allclose(gx[b, :se, :te + 1].sum(axis=1, dtype=np.float64), 1.0, rtol=1e-3)
        assert not gx[b, se:, :].any() and not gy[b, se + 1:, :].any()
        assert not gx[b, :, te + 1:].any() and not gy[b, :, te:].any()


@pytest.mark.parametrize("cfg", [(32, 200, 1000, False, 40), (16, 200, 1000, True, 25), (5, 300, 40, False, 40), (4, 1000, 3000, False, 10)])
def test_mi_launches_are_bit_reproducible(ft, dev, cfg):
    """The band hand-off (granules, polls) finishes at different times from launch to launch; the results must not care:
    repeated launches on the same inputs are bit-identical (a race in the hand-off, in the tiles or at the cut would
    show as a mismatch or a NaN).  scripts/mi_stress.py runs the same check 840 times."""
    from tf_fast_rnnt.mutual_information import mi_forward_backward
    B, S, T, mod, iters = cfg
    g = torch.Generator(device="cpu").manual_seed(S + T)
    px = (torch.randn((B, S, T if mod else T + 1), generator=g) - 6.0).to(dev)
    py = (torch.randn((B, S + 1, T), generator=g) - 6.0).to(dev)
    bd = torch.zeros((B, 4), dtype=torch.int32); bd[:, 2] = S; bd[:, 3] = T
    bd[1, 2] = S // 2; bd[1, 3] = T // 2 + 1
    bd = bd.to(dev)
    ref = None
    for _ in range(iters):
        out = mi_forward_backward(px, py, bd, True)
        if ref is None:
            ref = [o.clone() for o in out]
            assert torch.isfinite(ref[0]).all()
        else:
            assert all(torch.equal(a, b) for a, b in zip(out, ref))


def _raw_ws_calls(ft, dev, px, py, bd, ws, flags, modified=0, lib=None):
    """forward + backward through the _ws entry points on a caller-managed workspace."""
    from tf_fast_rnnt import _lib
    L = lib or _lib.lib()
    B, S, T1 = px.shape
    T = py.shape[2]
    st = torch.cuda.current_stream(dev).cuda_stream
    ans = torch.empty(B, device=dev); gx = torch.empty_like(px); gy = torch.empty_like(py)
    ag = torch.ones(B, device=dev)
    rc = L.ftr_mutual_information_fwd_ws_f32(px.data_ptr(), py.data_ptr(), bd.data_ptr(), ws.data_ptr(), ws.numel(), flags,
                                             ans.data_ptr(), B, S, T, modified, st)
    assert rc == 1, L.ftr_last_error()
    rc = L.ftr_mutual_information_bwd_ws_f32(px.data_ptr(), py.data_ptr(), bd.data_ptr(), ws.data_ptr(), ws.numel(), flags,
                                             None, gx.data_ptr(), gy.data_ptr(), ag.data_ptr(), 1, B, S, T, modified, st)
    assert rc == 1, L.ftr_last_error()
    return ans, gx, gy, ag


def _status(ft, dev, ws, B, S, T, lib=None):
    import ctypes
    from tf_fast_rnnt import _lib
    L = lib or _lib.lib()
    stt = ctypes.c_int(-1); dirty = ctypes.c_longlong(-1)
    rc = L.ftr_mutual_information_status(ws.data_ptr(), ws.numel(), B, S, T, ctypes.byref(stt), ctypes.byref(dirty),
                                         torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 1
    return stt.value, dirty.value


@pytest.mark.parametrize("modified", [0, 1])
def test_workspace_stays_clean_across_launches(ft, dev, modified):
    """The hand-off region is left all-zero by every launch (each consumer clears what it imported, the last band of an
    utterance clears its counters), whatever the boundaries were: one workspace initialised once and then used with
    FTR_MI_WS_CLEAN for a sequence of DIFFERENT boundaries gives bit-identical results to fresh, memset-per-call runs."""
    from tf_fast_rnnt import _lib
    L = _lib.lib()
    B, S, T = 6, 200, 300
    g = torch.Generator(device="cpu").manual_seed(1)
    px = (torch.randn((B, S, T if modified else T + 1), generator=g) - 4.0).to(dev)
    py = (torch.randn((B, S + 1, T), generator=g) - 4.0).to(dev)
    n = L.ftr_mutual_information_workspace_floats(B, S, T)
    ws = torch.full((n,), float("nan"), device=dev)        # garbage everywhere outside the region init clears
    st = torch.cuda.current_stream(dev).cuda_stream
    assert L.ftr_mutual_information_workspace_init(ws.data_ptr(), n, B, S, T, st) == 1
    rng = np.random.default_rng(0)
    for it in range(6):
        bd = torch.zeros((B, 4), dtype=torch.int32)
        bd[:, 2] = torch.from_numpy(rng.integers(0, S + 1, B).astype(np.int32))
        bd[:, 3] = torch.from_numpy(rng.integers(1, T + 1, B).astype(np.int32))
        if it == 0:
            bd[:, 2] = S; bd[:, 3] = T
        bd[0, 0] = min(3, int(bd[0, 2])); bd[0, 1] = min(5, int(bd[0, 3]))
        bd = bd.to(dev)
        got = _raw_ws_calls(ft, dev, px, py, bd, ws, _lib.FTR_MI_WS_CLEAN, modified)
        stt, dirty = _status(ft, dev, ws, B, S, T)
        assert stt == 0 and dirty == 0, (it, stt, dirty)
        fresh = torch.full((n,), float("nan"), device=dev)
        want = _raw_ws_calls(ft, dev, px, py, bd, fresh, 0, modified)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


def test_one_workspace_serves_a_ragged_loop(ft, dev):
    """A training loop pads every batch to its own maximum: 20 steps with 20 different (B, S, T).  The cached workspace is
    sized by capacity (hand-off region anchored at the end of the buffer, include/ftr.h FTR_MI_WS_CLEAN), so the loop
    allocates / zeroes when the capacity has to GROW only -- once when the largest shape comes first -- and every step's
    results are bit-identical to the same launch on a fresh, exactly-sized, freshly initialised workspace."""
    from tf_fast_rnnt import mutual_information as M
    from tf_fast_rnnt.mutual_information import mi_forward_backward
    rng = np.random.default_rng(11)
    shapes = [(6, 150, 400)] + [(int(rng.integers(1, 7)), int(rng.integers(1, 151)), int(rng.integers(2, 401))) for _ in range(19)]
    L = ft._lib.lib()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    M.clear_workspace_cache()
    inits0 = M._Workspace.inits
    for i, (B, S, T) in enumerate(shapes):
        px, py, bd = random_lattice(1000 + i, B, S, T, modified=bool(i % 2), ragged=True, begin_offsets=bool(i % 3 == 0))
        ans, gx, gy = mi_forward_backward(t(px), t(py), t(bd), True)
        # the same launches on a private workspace of exactly this shape's size
        nws = L.ftr_mutual_information_workspace_floats(B, S, T)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        a2, gx2, gy2, _ = _raw_ws_calls(ft, dev, t(px), t(py), t(bd), ws, 0, modified=i % 2)
        torch.cuda.synchronize()
        assert np.array_equal(ans.cpu().numpy(), a2.cpu().numpy(), equal_nan=True), (i, B, S, T)
        assert torch.equal(gx, gx2) and torch.equal(gy, gy2), (i, B, S, T)
    assert M._Workspace.inits - inits0 == 1, M._Workspace.inits - inits0
    assert M.check_workspace_status() == 0
    # growth: a larger shape reallocates once, after which the earlier shapes still fit
    px, py, bd = random_lattice(7, 8, 200, 500, ragged=True)
    mi_forward_backward(t(px), t(py), t(bd), True)
    px, py, bd = random_lattice(8, 2, 30, 50, ragged=True)
    mi_forward_backward(t(px), t(py), t(bd), True)
    assert M._Workspace.inits - inits0 == 2
    M.clear_workspace_cache()


def test_undersized_workspace_is_refused(ft, dev):
    """A buffer of the reference's p shape [B,S+1,T+1] is too small for the two-lattice workspace: the _ws entry points
    return FTR_ERR_INVALID_ARG instead of writing past its end."""
    from tf_fast_rnnt import _lib
    L = _lib.lib()
    B, S, T = 2, 30, 40
    px = torch.zeros((B, S, T + 1), device=dev); py = torch.zeros((B, S + 1, T), device=dev)
    small = torch.empty(B * (S + 1) * (T + 1), device=dev)
    ans = torch.empty(B, device=dev)
    rc = L.ftr_mutual_information_fwd_ws_f32(px.data_ptr(), py.data_ptr(), None, small.data_ptr(), small.numel(), 0,
                                             ans.data_ptr(), B, S, T, 0, torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0 and b"workspace" in L.ftr_last_error()


@pytest.mark.parametrize("cfg", [(160, 200, 300, False), (64, 1000, 2000, False), (300, 130, 90, True)])
def test_oversubscribed_grid(ft, dev, oracle, cfg):
    """More workgroups than the chip holds at once (1280 / 2048 / 1800 here; one workgroup per CU while the grid fits,
    a few per CU beyond): bands still only wait for bands with lower block ids, which were dispatched earlier.  Checked
    against the oracle on a few utterances and through the occupancy invariants on all."""
    from tf_fast_rnnt.mutual_information import mi_forward_backward
    B, S, T, mod = cfg
    g = torch.Generator(device="cpu").manual_seed(B)
    px = (torch.randn((B, S, T if mod else T + 1), generator=g) - 5.0)
    py = (torch.randn((B, S + 1, T), generator=g) - 5.0)
    bd = torch.zeros((B, 4), dtype=torch.int32); bd[:, 2] = S; bd[:, 3] = T
    bd[1, 2] = S // 2; bd[1, 3] = T // 2 + 1
    if mod:
        bd[:, 2] = torch.minimum(bd[:, 2], bd[:, 3])
    ans, gx, gy, chk = mi_forward_backward(px.to(dev), py.to(dev), bd.to(dev), True, return_ans_grad_check=True)
    torch.cuda.synchronize()
    assert torch.isfinite(ans).all()
    np.testing.assert_allclose(chk.cpu().numpy(), 1.0, rtol=1e-3)
    gyn = gy.cpu().numpy(); gxn = gx.cpu().numpy()
    for b in range(B):     # every frame is left exactly once: through a py transition, or (modified) through a px one
        te, se = int(bd[b, 3]), int(bd[b, 2])
        left = gyn[b, :se + 1, :te].sum(axis=0) + (gxn[b, :se, :te].sum(axis=0) if mod else 0.0)
        np.testing.assert_allclose(left, 1.0, rtol=1e-3)
    sel = [0, 1, B - 1]
    o_ans, o_p = oracle.mi_forward(px[sel].numpy(), py[sel].numpy(), bd[sel].numpy(), dtype=np.float64)
    o_gx, o_gy, _ = oracle.mi_backward(px[sel].numpy(), py[sel].numpy(), bd[sel].numpy(), o_p, dtype=np.float64)
    np.testing.assert_allclose(ans.cpu().numpy()[sel], o_ans, rtol=1e-4)
    assert max_rel(gx.cpu().numpy()[sel], o_gx.astype(np.float32)) <= 2e-3 and max_rel(gyn[sel], o_gy.astype(np.float32)) <= 2e-3


def test_nan_inputs_give_nan_ans(ft, dev, oracle):
    """Documented deviation (include/ftr.h): a NaN anywhere among the px / py entries inside the boundary rectangle of an
    utterance gives ans = NaN for that utterance (the reference's LogAdd drops or keeps a NaN depending on the argument
    it arrives in); the other utterances of the batch are unaffected."""
    from tf_fast_rnnt.mutual_information import mi_forward_backward
    px, py, bd = random_lattice(9, 5, 140, 180, ragged=False)
    px[1, 70, 90] = np.nan
    py[3, 139, 3] = np.nan
    py[4, 10, 179] = np.nan      # inside the rectangle (t < t_end)
    px[2, 5, 180] = np.nan       # column t == T of px is inside the rectangle as well (regular type)
    ans, gx, gy = mi_forward_backward(torch.from_numpy(px).to(dev), torch.from_numpy(py).to(dev), torch.from_numpy(bd).to(dev), True)
    torch.cuda.synchronize()
    a = ans.cpu().numpy()
    assert np.isnan(a[[1, 2, 3, 4]]).all() and np.isfinite(a[0])
    o_ans, _ = oracle.mi_forward(px[:1], py[:1], bd[:1])
    np.testing.assert_allclose(a[0], o_ans[0], rtol=1e-4)
    assert not torch.isnan(gx[0]).any() and not torch.isnan(gy[0]).any()
    # a second launch on the (cached, self-cleaned) workspace with clean inputs: no flag survives
    px2, py2, _ = random_lattice(10, 5, 140, 180, ragged=False)
    ans2, _, _ = mi_forward_backward(torch.from_numpy(px2).to(dev), torch.from_numpy(py2).to(dev), torch.from_numpy(bd).to(dev), True)
    assert torch.isfinite(ans2).all()


def test_absent_producer_poisons_loudly(ft, dev):
    """Test build of the library in which the first alpha band never publishes its hand-off granules
    (csrc/_build/libftr_nopublish.so, FTR_MAX_SPIN lowered): the consumer's bounded poll gives up, the launch terminates,
    the sticky status word is set and every ans is NaN -- a stalled producer cannot go unnoticed."""
    import os
    import diag
    L = diag.load(os.path.join(diag.BUILD, "libftr_nopublish.so"))
    B, S, T = 3, 150, 200          # 3 bands per direction
    g = torch.Generator(device="cpu").manual_seed(2)
    px = (torch.randn((B, S, T + 1), generator=g) - 4.0).to(dev); py = (torch.randn((B, S + 1, T), generator=g) - 4.0).to(dev)
    bd = torch.tensor([[0, 0, S, T]] * B, dtype=torch.int32, device=dev)
    ws = torch.empty(L.ftr_mutual_information_workspace_floats(B, S, T), device=dev)
    ans, gx, gy, _ = _raw_ws_calls(ft, dev, px, py, bd, ws, 0, lib=L)
    torch.cuda.synchronize()
    stt, _ = _status(ft, dev, ws, B, S, T, lib=L)
    assert stt == 1
    assert torch.isnan(ans).all()


def _band_case(ft, dev, B, T, S, r, modified, seed, offsets=False, break_end=False):
    """Random band arrays on hand-built monotone ranges -> (band kernel results, lattice kernel results mapped to the band)."""
    from tf_fast_rnnt import _lib
    rng = np.random.default_rng(seed)
    s0 = np.zeros((B, T), np.int64)
    for b in range(B):
        steps = rng.integers(0, max(r, 1), T)                          # 0 <= s0[t+1] - s0[t] <= r - 1
        s0[b] = np.minimum(np.cumsum(steps) - steps[0], max(S + 1 - r, 0))
        s0[b] = np.maximum.accumulate(s0[b])
    if break_end:
        s0[:, :] = 0                                                   # the band never reaches row S: no path (for S >= r)
    ranges = (s0[:, :, None] + np.arange(r)[None, None, :]).astype(np.int32)
    pxb = (-rng.random((B, T, r)) * 3 - 0.05).astype(np.float32)
    pyb = (-rng.random((B, T, r)) * 3 - 0.05).astype(np.float32)
    bd = np.zeros((B, 4), np.int32); bd[:, 2] = S; bd[:, 3] = T
    if offsets:
        for b in range(B):
            tb = int(rng.integers(0, max(T // 3, 1))); te = int(rng.integers(max(tb + 1, (2 * T) // 3), T + 1))
            lo, hi = int(s0[b, tb]), int(min(s0[b, te - 1] + r - 1, S))
            sb = int(rng.integers(lo, min(lo + r, hi + 1))); se = int(rng.integers(max(sb, hi - r + 1), hi + 1))
            bd[b] = (sb, tb, se, te)
    T1 = T if modified else T + 1
    px = np.full((B, S, T1), -np.inf, np.float32); py = np.full((B, S + 1, T), -np.inf, np.float32)
    for b in range(B):
        for t in range(T):
            for k in range(r):
                s = s0[b, t] + k
                if s < S: px[b, s, t] = pxb[b, t, k]
                if s <= S: py[b, s, t] = pyb[b, t, k]
                if s >= S: pxb[b, t, k] = -np.inf                     # what the band builder writes there
                if s > S: pyb[b, t, k] = -np.inf
    if not modified:                                                   # fix_for_boundary (rnnt_loss.py:28-61): no symbol in column t_end
        for b in range(B):
            te = int(bd[b, 3])
            if te < T: px[b, :, te] = -np.inf; pxb[b, te, :] = -np.inf
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tpx, tpy, tbd, trg = t_(pxb), t_(pyb), t_(bd), t_(ranges)
    ans = torch.empty(B, device=dev); gxb = torch.empty_like(tpx); gyb = torch.empty_like(tpy)
    st = torch.cuda.current_stream().cuda_stream
    assert _lib.lib().ftr_mutual_information_band_supported(T, S, r) in (1, 2)    # 2: the streaming kernel (long utterances)
    nws = _lib.lib().ftr_mutual_information_band_workspace_floats(B, T, S, r)
    bws = torch.empty(max(nws, 1), device=dev)
    _lib.call("ftr_mutual_information_band_ws_f32", tpx.data_ptr(), tpy.data_ptr(), trg.data_ptr(), tbd.data_ptr(), bws.data_ptr(), nws,
              ans.data_ptr(), gxb.data_ptr(), gyb.data_ptr(), B, T, S, r, int(modified), st)
    lpx = t_(px).requires_grad_(True); lpy = t_(py).requires_grad_(True)
    lans = ft.mutual_information_recursion(lpx, lpy, tbd)
    fin = torch.isfinite(lans)
    if fin.any(): lans[fin].sum().backward()
    lgx = np.zeros_like(px) if lpx.grad is None else lpx.grad.cpu().numpy()
    lgy = np.zeros_like(py) if lpy.grad is None else lpy.grad.cpu().numpy()
    egx = np.zeros((B, T, r), np.float32); egy = np.zeros((B, T, r), np.float32)
    for b in range(B):
        for t in range(T):
            for k in range(r):
                s = s0[b, t] + k
                if s < S and t < T1: egx[b, t, k] = lgx[b, s, t]
                if s <= S: egy[b, t, k] = lgy[b, s, t]
    return ans.cpu().numpy(), gxb.cpu().numpy(), gyb.cpu().numpy(), lans.detach().cpu().numpy(), egx, egy, fin.cpu().numpy()


@pytest.mark.parametrize("modified", [False, True])
@pytest.mark.parametrize("case", [dict(B=3, T=37, S=11, r=4, offsets=True), dict(B=2, T=1, S=0, r=1), dict(B=2, T=9, S=0, r=1),
                                  dict(B=2, T=24, S=9, r=1), dict(B=2, T=50, S=30, r=15), dict(B=2, T=50, S=30, r=9, offsets=True),
                                  dict(B=2, T=40, S=20, r=3, break_end=True), dict(B=2, T=6, S=40, r=8), dict(B=1, T=300, S=100, r=8, offsets=True),
                                  dict(B=2, T=2200, S=500, r=5), dict(B=1, T=1500, S=700, r=10, offsets=True)])
@pytest.mark.parametrize("impl", ["chain", "segments"])
def test_band_recursion_kernel_edge_cases(ft, dev, case, modified, impl, monkeypatch):
    """ftr_mutual_information_band_ws_f32 directly against the full-lattice kernels on the lattices the band expands to:
    begin / end offsets inside the band, S = 0, T = 1, one-row bands, 16-lane chains (r > 8), bands that never reach the end
    cell (ans = -inf on both routes, zero occupancies), lattices taller than long; the last two are too long for LDS (the
    streaming chain kernel, workspace in global memory).  Both implementations of the band recursion on every case: the chain
    kernels of mi_band.hip and the segmented route of mi_band_seg.hip (which the library itself takes from S + T >= 1100)."""
    monkeypatch.setenv("FTR_BAND_IMPL", impl)
    a, gx, gy, la, egx, egy, fin = _band_case(ft, dev, modified=modified, seed=5, **case)
    assert np.array_equal(np.isfinite(a), fin)
    if fin.any():
        np.testing.assert_allclose(a[fin], la[fin], rtol=1e-5, atol=1e-5)
        tol = 2e-5 if case["T"] <= 500 else 1e-4      # two float32 evaluations; the long cases accumulate more rounding
        assert np.abs(gx[fin] - egx[fin]).max() <= tol and np.abs(gy[fin] - egy[fin]).max() <= tol
    assert np.isfinite(gx).all() and np.isfinite(gy).all()
    assert (gx[~fin] == 0).all() and (gy[~fin] == 0).all()


def test_band_recursion_rejects_non_monotone_ranges(ft, dev):
    """ranges[b,t,0] decreasing somewhere inside the rectangle: the band kernel answers NaN / zero occupancies instead of
    computing on colliding slots (include/ftr.h, precondition of ftr_mutual_information_band_f32)."""
    from tf_fast_rnnt import _lib
    B, T, S, r = 2, 20, 8, 3
    s0 = np.minimum(np.arange(T) // 3, S + 1 - r)
    ranges = np.tile((s0[:, None] + np.arange(r)[None, :])[None], (B, 1, 1)).astype(np.int32)
    ranges[1, 10] -= 1                                   # utterance 1: one step backwards
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pxb = t_(-np.ones((B, T, r), np.float32)); pyb = t_(-np.ones((B, T, r), np.float32))
    bd = t_(np.array([[0, 0, S, T]] * B, np.int32))
    ans = torch.empty(B, device=dev); gx = torch.full((B, T, r), 7.0, device=dev); gy = torch.full((B, T, r), 7.0, device=dev)
    _lib.call("ftr_mutual_information_band_f32", pxb.data_ptr(), pyb.data_ptr(), t_(ranges).data_ptr(), bd.data_ptr(), ans.data_ptr(),
              gx.data_ptr(), gy.data_ptr(), B, T, S, r, 0, torch.cuda.current_stream().cuda_stream)
    a = ans.cpu().numpy()
    assert np.isfinite(a[0]) and np.isnan(a[1])
    assert (gx[1] == 0).all() and (gy[1] == 0).all() and float(gx[0].sum() + gy[0].sum()) > 0


def test_recursion_fuzz_against_plain_kernels(ft, dev):
    """scripts/mi_fuzz.py: 150 random lattices (shapes around the band and chunk sizes, begin / end offsets, -inf entries, both
    types): the wavefront kernels against the plain one-thread-per-row kernels."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("mi_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "mi_fuzz.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.main(150, 20261004)


def test_streaming_band_kernel_on_every_size(dev):
    """The streaming band kernel (long utterances: arrays in a global workspace) on the SMALL cases too: FTR_BAND_FORCE_STREAM
    (read once per process, hence child processes) sends every size through it -- the edge-case test above and the
    route-vs-route fuzz of scripts/band_fuzz.py."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FTR_BAND_FORCE_STREAM="1")
    r1 = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_mi.py"), "-q", "-x", "-m", "gpu", "-k",
                         "band_recursion_kernel_edge_cases", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, os.path.join(root, "scripts", "band_fuzz.py"), "150", "31"], env=env, cwd=root, capture_output=True,
                        text=True, timeout=600)
    assert r2.returncode == 0 and "band route == lattice route" in r2.stdout, r2.stdout[-2000:] + r2.stderr[-2000:]


def test_segmented_band_route_on_every_size(dev):
    """The segmented band recursion (mi_band_seg.hip: transfer matrices per segment, float64 chains, occupancies as
    exp(p + q - ans)) forced onto every size by FTR_BAND_IMPL=segments -- the library itself takes it from S + T >= 1100 --
    through the whole pruned loss: the route-vs-route fuzz of scripts/band_fuzz.py (random shapes down to T = 1, both types,
    ragged boundaries, r up to 20) against the full-lattice route."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FTR_BAND_IMPL="segments")
    r2 = subprocess.run([sys.executable, os.path.join(root, "scripts", "band_fuzz.py"), "150", "77"], env=env, cwd=root, capture_output=True,
                        text=True, timeout=600)
    assert r2.returncode == 0 and "band route == lattice route" in r2.stdout, r2.stdout[-2000:] + r2.stderr[-2000:]


@pytest.mark.parametrize("B", [1, 3, 64, 300])
def test_loss_tail_inside_the_backward_launch(ft, dev, B):
    """ftr_mutual_information_bwd_loss_ws_f32 = the backward launch (seed of ones) + the loss tail: the occupancies of
    ftr_mutual_information_bwd_ws_f32 and -ans / -mean / -sum as ftr_negated_reduce_f32 computes them, bit for bit."""
    from tf_fast_rnnt import _lib
    from tf_fast_rnnt.mutual_information import mi_forward_backward
    g = torch.Generator(device="cpu").manual_seed(B)
    S, T = 9, 21
    px = (torch.randn((B, S, T + 1), generator=g) - 2.0).to(dev); py = (torch.randn((B, S + 1, T), generator=g) - 1.0).to(dev)
    ans, gx, gy = mi_forward_backward(px, py, None, True, ans_grad_is_one=True)
    st = torch.cuda.current_stream().cuda_stream
    for code in (0, 1, 2):
        want = torch.empty((B,) if code == 0 else (), device=dev)
        _lib.call("ftr_negated_reduce_f32", ans.data_ptr(), B, code, want.data_ptr(), st)
        ans2, gx2, gy2, loss = mi_forward_backward(px, py, None, True, ans_grad_is_one=True, loss_code=code)
        assert torch.equal(ans2, ans) and torch.equal(gx2, gx) and torch.equal(gy2, gy)
        assert loss.shape == want.shape and torch.equal(loss, want), (code, loss, want)
