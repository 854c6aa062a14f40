"""Shared input recipes for the parity tests (no reference code: the recipe restates the shapes and
distributions of tf_fast_rnnt/python/tests/simple_rnnt_loss_test.py:51-66 and SURVEY.md 8d)."""
import numpy as np


def reference_test_recipe(seed, B, T, S, C):
    """simple_rnnt_loss_test.py:260-289: legacy numpy seeding, ragged frames / seq_length."""
    rs = np.random.RandomState(seed)
    frames = rs.randint(S, T, (B,))
    seq_length = rs.randint(3, S - 1, (B,))
    T = int(np.amax(frames)); S = int(np.amax(seq_length))
    am = rs.randn(B, T, C).astype("f")
    lm = rs.randn(B, S + 1, C).astype("f")
    symbols = rs.randint(0, C - 1, (B, S)).astype(np.int32)
    boundary = np.zeros((B, 4), dtype=np.int32)
    boundary[:, 2] = seq_length
    boundary[:, 3] = frames
    return dict(am=am, lm=lm, symbols=symbols, boundary=boundary, termination_symbol=C - 1, B=B, T=T, S=S, C=C)


def synthetic(seed, B, T, S, C, ragged=False):
    """SURVEY.md 8d synthetic inputs."""
    rng = np.random.default_rng(seed)
    am = rng.standard_normal((B, T, C)).astype(np.float32)
    lm = rng.standard_normal((B, S + 1, C)).astype(np.float32)
    symbols = rng.integers(0, C - 1, (B, S)).astype(np.int32)
    boundary = np.zeros((B, 4), dtype=np.int32)
    if ragged:
        t_end = rng.integers((T + 1) // 2, T + 1, (B,))
        t_end[0] = T
        s_hi = np.minimum(S, t_end)
        s_end = np.array([rng.integers(min((S + 1) // 2, hi), hi + 1) for hi in s_hi])
        s_end[0] = S
    else:
        t_end = np.full((B,), T); s_end = np.full((B,), S)
    boundary[:, 2] = s_end
    boundary[:, 3] = t_end
    return dict(am=am, lm=lm, symbols=symbols, boundary=boundary, termination_symbol=C - 1, B=B, T=T, S=S, C=C)


def random_lattice(seed, B, S, T, modified=False, neg_inf_frac=0.0, ragged=True, begin_offsets=False):
    """px/py drawn directly (for the native-op tests), optional -inf entries and ragged boundaries."""
    rng = np.random.default_rng(seed)
    T1 = T if modified else T + 1
    px = (rng.standard_normal((B, S, T1)) - 1.0).astype(np.float32)
    py = (rng.standard_normal((B, S + 1, T)) - 1.0).astype(np.float32)
    if neg_inf_frac > 0:
        px[rng.random(px.shape) < neg_inf_frac] = -np.inf
        py[rng.random(py.shape) < neg_inf_frac] = -np.inf
    bd = np.zeros((B, 4), dtype=np.int32)
    bd[:, 2] = S; bd[:, 3] = T
    if ragged and B > 1:
        for b in range(1, B):
            bd[b, 2] = rng.integers(0, S + 1)
            bd[b, 3] = rng.integers(max(1, T // 3), T + 1)
            if begin_offsets:
                bd[b, 0] = rng.integers(0, bd[b, 2] + 1)
                bd[b, 1] = rng.integers(0, bd[b, 3] + 1)
    return px, py, bd


def max_rel(a, b):
    """normwise relative error: max|a-b| / max(|b|, tiny)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), "finite / non-finite pattern differs"
    assert np.array_equal(a[~fin], b[~fin]) or (np.isnan(a[~fin]) == np.isnan(b[~fin])).all()
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(a[fin] - b[fin])) / max(np.max(np.abs(b[fin])), 1e-30))


def assert_parity(got, ref32, ref64=None, tol=1e-4, what=""):
    """north_star tolerance: within `tol` (normwise relative) of the float32 oracle (= the reference's
    arithmetic).  Where float32 log-domain arithmetic is itself less accurate than `tol` (long lattices: the
    float32 oracle is 3e-4 .. 7e-3 away from the float64 oracle), the native result must instead be
    at least as close to the float64 oracle as the float32 oracle is."""
    e32 = max_rel(got, ref32)
    if e32 <= tol:
        return e32
    assert ref64 is not None, f"{what}: {e32:.3g} > {tol} vs float32 oracle and no float64 reference given"
    e64 = max_rel(got, ref64)
    eref = max_rel(ref32, ref64)
    assert e64 <= max(tol, eref), f"{what}: vs f32 oracle {e32:.3g}, vs f64 oracle {e64:.3g}, f32 oracle vs f64 oracle {eref:.3g}"
    return e64
