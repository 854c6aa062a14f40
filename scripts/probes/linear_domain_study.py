"""Numerics study for a candidate next step of the lattice recursion: carry the chain in a scaled
LINEAR domain -- per row (lane) a float L and a block exponent R, p = R + log2 L, L <- ldexp(L_up, R_up - R) * 2^X + L * 2^Y,
renormalised (frexp) every K wavefront steps -- instead of log2-domain logadd per cell.  chain_probe.hip measures what that
chain costs on the GPU (11 ns per step against 26 ns); this script measures what it does to the results, on CPU, in numpy:
float64 log-domain recursion = truth, float32 log-domain (the arithmetic the shipped kernel uses) and the candidate beside it.
python scripts/probes/linear_domain_study.py        (self-contained: no oracle, no GPU)"""
import numpy as np

LOG2E = np.float64(1.4426950408889634)

def ref_logdomain(px, py, dtype):
    """p over the (S+1) x (T+1) lattice, by anti-diagonals; px[s,t]: (s,t)->(s+1,t), py[s,t]: (s,t)->(s,t+1)."""
    S, T = px.shape[0], py.shape[1]
    px = px.astype(dtype); py = py.astype(dtype)
    p = np.full((S + 1, T + 1), -np.inf, dtype=dtype); p[0, 0] = 0
    for j in range(1, S + T + 1):
        s = np.arange(max(0, j - T), min(S, j) + 1); t = j - s
        a = np.full(s.shape, -np.inf, dtype=dtype); b = a.copy()
        m = s > 0; a[m] = p[s[m] - 1, t[m]] + px[s[m] - 1, t[m]]
        m = t > 0; b[m] = p[s[m], t[m] - 1] + py[s[m], t[m] - 1]
        with np.errstate(invalid="ignore", divide="ignore"):
            mx = np.maximum(a, b); d = -np.abs(a - b)
            v = mx + np.log1p(np.exp(d.astype(dtype))).astype(dtype)
        v[np.isneginf(mx)] = -np.inf
        p[s, t] = v
    return p

def backward_q(px, py):
    S, T = px.shape[0], py.shape[1]
    q = np.full((S + 1, T + 1), -np.inf); q[S, T] = 0
    for j in range(S + T - 1, -1, -1):
        s = np.arange(max(0, j - T), min(S, j) + 1); t = j - s
        a = np.full(s.shape, -np.inf); b = a.copy()
        m = s < S; a[m] = q[s[m] + 1, t[m]] + px[s[m], t[m]]
        m = t < T; b[m] = q[s[m], t[m] + 1] + py[s[m], t[m]]
        with np.errstate(invalid="ignore", divide="ignore"):
            mx = np.maximum(a, b); v = mx + np.log1p(np.exp(-np.abs(a - b)))
        v[np.isneginf(mx)] = -np.inf
        q[s, t] = v
    return q

def candidate(px, py, K):
    """float32 scaled-linear recursion, wavefront order (lane = row s, step j handles t = j - s), returns p in nats and the
    number of terms that were flushed although their true size was not negligible against the lane's own value (the event a
    kernel would have to flag for the log-domain fallback)."""
    S, T = px.shape[0], py.shape[1]
    f = np.float32
    with np.errstate(over="ignore", under="ignore"):
        EX = np.exp2((px.astype(np.float64) * LOG2E)).astype(f)      # prepared by the IO wave: 2^X, flushed below 2^-149
        EY = np.exp2((py.astype(np.float64) * LOG2E)).astype(f)
    L = np.zeros(S + 1, dtype=f); R = np.zeros(S + 1, dtype=np.int32)
    L[0] = 1
    p = np.full((S + 1, T + 1), -np.inf); p[0, 0] = 0
    flagged = 0
    for j in range(1, S + T + 1):
        s = np.arange(max(0, j - T), min(S, j) + 1); t = j - s
        Lup = np.zeros(s.shape, dtype=f); ex = np.zeros(s.shape, dtype=f); ey = np.zeros(s.shape, dtype=f)
        m = s > 0
        dR = np.zeros(s.shape, dtype=np.int64); dR[m] = R[s[m] - 1].astype(np.int64) - R[s[m]]
        with np.errstate(over="ignore", under="ignore"):
            Lup[m] = np.ldexp(L[s[m] - 1], np.clip(dR[m], -300, 300)).astype(f)
        ex[m] = EX[s[m] - 1, t[m]]
        mt = t > 0; ey[mt] = EY[s[mt], t[mt] - 1]
        own = np.where(mt, L[s], f(0))
        # a lane that holds nothing yet (L = 0) adopts the exponent of what arrives from above: R <- R_up, L_up unscaled
        empty = (own == 0) & m
        if empty.any():
            R[s[empty]] = R[s[empty] - 1]; Lup[empty] = L[s[empty] - 1]
        with np.errstate(over="ignore", under="ignore", invalid="ignore"):
            new = (Lup * ex + own * ey).astype(f)
        flagged += int(np.sum(~np.isfinite(new)))
        L[s] = new
        if j % K == 0:       # renormalise every K steps, all lanes at once
            mant, e = np.frexp(L[s]); nz = L[s] != 0
            L[s] = np.where(nz, mant, 0).astype(f); R[s] = np.where(nz, R[s] + e, R[s])
        with np.errstate(divide="ignore"):
            p[s, t] = (R[s] + np.log2(L[s].astype(np.float64))) / LOG2E
    return p, flagged

def cases(rng, S=100, T=400, C=500):
    out = {}
    out["synthetic randn-6 (bench.py's inputs)"] = (rng.standard_normal((S, T + 1)) - 6, rng.standard_normal((S + 1, T)) - 6)
    # a sharp model: on a random monotone path the right move has probability ~0.9, everything else is far down
    for name, off in (("sharp model, off-path moves at -25 nats", -25.0), ("very sharp model, off-path moves at -60 nats", -60.0),
                      ("degenerate model, off-path moves at -120 nats", -120.0)):
        px = off + rng.standard_normal((S, T + 1)); py = off + rng.standard_normal((S + 1, T))
        s = 0
        emit_at = np.sort(rng.choice(T, S, replace=False))
        for t in range(T):
            if s < S and t == emit_at[s]: px[s, t] = -0.1; s += 1
            py[s, t] = -0.1
        out[name] = (px, py)
    # every path must cross improbable emissions: symbol 7 has probability e^-100 at every frame
    px = rng.standard_normal((S, T + 1)) - 6; py = rng.standard_normal((S + 1, T)) - 1; px[7, :] = -100.0; px[40, :] = -95.0
    out["two symbols at e^-100 on every frame (forced crossing)"] = (px, py)
    return out

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for name, (px, py) in cases(rng).items():
        px[:, -1] = -np.inf       # no emission after the last frame (regular type)
        p64 = ref_logdomain(px, py, np.float64); q = backward_q(px, py); total = p64[-1, -1]
        occ = np.exp(p64 + q - total); rel = occ > 1e-6     # the cells that carry the loss and its gradient
        p32 = ref_logdomain(px, py, np.float32).astype(np.float64)
        print(f"{name}: loss {-total:.3f} nats, {int(rel.sum())} relevant cells")
        print(f"   float32 log domain (shipped): |ans err| {abs(p32[-1, -1] - total):.2e}   max |p err| on relevant cells {np.abs(p32 - p64)[rel].max():.2e}")
        for K in (4, 8, 16):
            pc, flagged = candidate(px, py, K)
            with np.errstate(invalid="ignore"):
                err = np.abs(pc - p64)
            lost = int(np.sum(np.isneginf(pc) & np.isfinite(p64)))
            lost_rel = int(np.sum(np.isneginf(pc) & rel))
            e_rel = err[rel & np.isfinite(pc)].max() if (rel & np.isfinite(pc)).any() else float("nan")
            print(f"   scaled linear, K={K:2d}: |ans err| {abs(pc[-1, -1] - total):.2e}   max |p err| on relevant cells {e_rel:.2e}   "
                  f"cells flushed to -inf {lost} (relevant: {lost_rel})   overflow events {flagged}")
