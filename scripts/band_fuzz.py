"""Randomised comparison of the two routes of rnnt_loss_pruned: the band-native recursion (ranges straight from
get_rnnt_prune_ranges, or a clone of them) against the full-lattice route (FTR_PRUNED_ROUTE=lattice).
python scripts/band_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tf-fast-rnnt_amd", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import tf_fast_rnnt as ft
from helpers import synthetic


def main(n=100, seed=0):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst_l = worst_g = 0.0; band = 0
    for it in range(n):
        B = int(rng.integers(1, 4)); T = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 150])); S = int(rng.choice([1, 2, 5, 9, 17, 40, 70]))
        C = int(rng.choice([4, 7, 8, 12, 33])); r = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 20]))
        rt = "modified" if rng.integers(0, 2) else "regular"
        if rt == "modified" and S > T: S = T
        d = synthetic(int(rng.integers(1, 10**6)), B, T, S, C, ragged=bool(rng.integers(0, 2)))
        blank = d["termination_symbol"]
        t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        am, lm, sym, bd = (t_(d[k]) for k in ("am", "lm", "symbols", "boundary"))
        _, (gx, gy) = ft.rnnt_loss_simple(lm, am, sym, blank, bd, rt, reduction="sum", calc_gradients=True)
        ranges = ft.get_rnnt_prune_ranges(gx, gy, bd, r)
        am_p, lm_p = ft.do_rnnt_pruning(am, lm, ranges)
        base = torch.tanh(am_p + lm_p).detach()
        outs = []
        for route in ("band", "lattice"):
            os.environ["FTR_PRUNED_ROUTE"] = route
            rg = ranges.clone() if it % 2 else ranges       # a clone carries no mark: it is checked on the device and takes the same route
            logits = base.clone().requires_grad_(True)
            loss = ft.rnnt_loss_pruned(logits, sym, rg, blank, bd, rt, 0.1, "none")
            fin = torch.isfinite(loss)
            if fin.any(): loss[fin].sum().backward()
            outs.append((loss.detach().cpu().numpy(), np.zeros(base.shape, np.float32) if logits.grad is None else logits.grad.cpu().numpy()))
        os.environ["FTR_PRUNED_ROUTE"] = "band"
        from tf_fast_rnnt.rnnt_loss import _band_path_ok
        band += int(_band_path_ok(ranges, bd, T, S, ranges.shape[2]))
        f0, f1 = np.isfinite(outs[0][0]), np.isfinite(outs[1][0])
        assert np.array_equal(f0, f1), (it, B, T, S, C, r, rt, outs[0][0], outs[1][0])
        if f0.any():
            el = np.abs(outs[0][0][f0] - outs[1][0][f0]).max() / max(1.0, np.abs(outs[1][0][f0]).max())
            eg = np.abs(outs[0][1][f0] - outs[1][1][f0]).max() / max(1e-6, np.abs(outs[1][1][f0]).max())
            worst_l, worst_g = max(worst_l, el), max(worst_g, eg)
            assert el < 1e-5 and eg < 5e-4, (it, B, T, S, C, r, rt, el, eg)
        assert np.isfinite(outs[0][1]).all()
    print(f"{n} random cases ({band} through the band kernel): band route == lattice route (worst loss {worst_l:.1e}, gradient {worst_g:.1e})")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
