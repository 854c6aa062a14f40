"""Randomised comparison of the fused builder kernels (csrc/simple_fused.hip: MFMA contraction + epilogue; opt-in fused d am)
with the library-GEMM route: px / py of the simple and smoothed builders, and the gradients of both losses w.r.t. am, lm.
python scripts/builder_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tf-fast-rnnt_amd", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import tf_fast_rnnt as ft
from helpers import synthetic


def main(n=60, seed=0, tol=1e-4):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst = 0.0
    for it in range(n):
        B = int(rng.integers(1, 4)); T = int(rng.choice([1, 2, 5, 16, 63, 64, 65, 100, 130])); S = int(rng.choice([0, 1, 3, 15, 16, 17, 60, 111, 120, 210]))
        C = 4 * int(rng.integers(1, 80))
        rt = "modified" if rng.integers(0, 2) else "regular"
        if rt == "modified" and S > T: S = T
        d = synthetic(int(rng.integers(1, 10**6)), B, T, S, C, ragged=bool(rng.integers(0, 2)))
        blank = d["termination_symbol"]
        t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        sym, bd = t_(d["symbols"]), t_(d["boundary"])
        pen = float(rng.choice([0.0, 0.25]))
        wts = torch.linspace(0.5, 1.5, B, device=dev)

        def run():
            out = []
            lm, am = t_(d["lm"]), t_(d["am"])
            out += [x.cpu().numpy() for x in ft.get_rnnt_logprobs(lm, am, sym, blank, rnnt_type=rt, boundary=bd)]
            out += [x.cpu().numpy() for x in ft.get_rnnt_logprobs_smoothed(lm, am, sym, blank, lm_only_scale=0.15, am_only_scale=0.1, boundary=bd, rnnt_type=rt)]
            for fn in (lambda l, a: ft.rnnt_loss_simple(l, a, sym, blank, boundary=bd, rnnt_type=rt, delay_penalty=pen, reduction="none"),
                       lambda l, a: ft.rnnt_loss_smoothed(l, a, sym, blank, lm_only_scale=0.15, am_only_scale=0.1, boundary=bd, rnnt_type=rt, delay_penalty=pen, reduction="none")):
                l = t_(d["lm"]).requires_grad_(True); a = t_(d["am"]).requires_grad_(True)
                loss = fn(l, a); fin = torch.isfinite(loss)
                if fin.any(): (loss[fin] * wts[fin]).sum().backward()
                out += [loss.detach().cpu().numpy()] + [np.zeros(x.shape, np.float32) if x.grad is None else x.grad.cpu().numpy() for x in (a, l)]
            return out

        os.environ["FTR_BUILDER_GEMM"] = "library"; os.environ["FTR_BUILDER_BWD"] = "library"
        ref = run()
        os.environ["FTR_BUILDER_GEMM"] = "fused"; os.environ["FTR_BUILDER_BWD"] = "fused"
        got = run()
        os.environ.pop("FTR_BUILDER_GEMM"); os.environ.pop("FTR_BUILDER_BWD")
        for k, (g, r_) in enumerate(zip(got, ref)):
            assert g.shape == r_.shape and np.array_equal(np.isneginf(g), np.isneginf(r_)) and np.array_equal(np.isposinf(g), np.isposinf(r_)), (it, B, T, S, C, rt, k)
            fin = np.isfinite(r_)
            if fin.any():
                e = np.abs(g[fin] - r_[fin]).max() / max(1.0, np.abs(r_[fin]).max())
                worst = max(worst, e)
                assert e <= tol and np.isfinite(g[fin]).all(), (it, B, T, S, C, rt, k, e)
    print(f"{n} random cases: fused builder kernels == library-GEMM route (worst normwise deviation {worst:.1e})")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
