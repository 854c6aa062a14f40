"""Randomised comparison of the wavefront recursion (product) with the plain one-thread-per-row kernels (reference
arithmetic on the device, from the test-only diag library: tests/diag.py): shapes, ragged boundaries with begin offsets, both types, -inf entries.
python scripts/mi_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tf_fast_rnnt as ft
import diag


def main(n=100, seed=0):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst = 0.0
    for it in range(n):
        B = int(rng.integers(1, 5)); S = int(rng.choice([0, 1, 2, 5, 17, 63, 64, 65, 100, 130, 200, 257])); T = int(rng.choice([1, 2, 3, 15, 16, 17, 40, 64, 100, 130, 257, 500]))
        mod = bool(rng.integers(0, 2))
        T1 = T if mod else T + 1
        px = (rng.standard_normal((B, S, T1)) - 2).astype(np.float32); py = (rng.standard_normal((B, S + 1, T)) - 2).astype(np.float32)
        if rng.random() < 0.3: px[rng.random(px.shape) < 0.05] = -np.inf
        if not mod: px[:, :, T] = -np.inf
        bd = np.zeros((B, 4), np.int32)
        for b in range(B):
            sb = int(rng.integers(0, S + 1)) if rng.random() < 0.5 else 0; tb = int(rng.integers(0, T + 1)) if rng.random() < 0.5 else 0
            se = int(rng.integers(sb, S + 1)); te = int(rng.integers(tb, T + 1))
            if rng.random() < 0.5: se, te = S, T
            bd[b] = (sb, tb, se, te)
        res = []
        for impl in (0, 1):
            tx = torch.from_numpy(px).to(dev).requires_grad_(True); ty = torch.from_numpy(py).to(dev).requires_grad_(True)
            if impl == 0:
                ans = ft.mutual_information_recursion(tx, ty, torch.from_numpy(bd).to(dev))
                fin = torch.isfinite(ans)
                if fin.any(): ans[fin].sum().backward()
                gx = np.zeros_like(px) if tx.grad is None else tx.grad.cpu().numpy(); gy = np.zeros_like(py) if ty.grad is None else ty.grad.cpu().numpy()
            else:   # the gradient of the sum over the finite answers = the backward seeded with their indicator
                ans = diag.plain_forward_backward(tx.detach(), ty.detach(), torch.from_numpy(bd).to(dev), False)[0]
                fin = torch.isfinite(ans)
                _, gxt, gyt, _ = diag.plain_forward_backward(tx.detach(), ty.detach(), torch.from_numpy(bd).to(dev), True, ans_grad=fin.float())
                gx, gy = gxt.cpu().numpy(), gyt.cpu().numpy()
            res.append((ans.detach().cpu().numpy(), gx, gy, fin.cpu().numpy()))
        (a0, x0, y0, f0), (a1, x1, y1, f1) = res
        assert np.array_equal(f0, f1), (it, B, S, T, mod, bd, a0, a1)
        if f0.any():
            ea = np.abs(a0[f0] - a1[f0]).max() / max(1.0, np.abs(a1[f0]).max())
            eg = max(np.abs(x0[f0] - x1[f0]).max() if S else 0.0, np.abs(y0[f0] - y1[f0]).max())
            worst = max(worst, ea, eg)
            assert ea < 1e-5 and eg < 5e-3, (it, B, S, T, mod, bd.tolist(), ea, eg)
        assert np.isfinite(x0).all() and np.isfinite(y0).all()
    print(f"{n} random cases: wavefront == plain (worst deviation {worst:.2e})")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
