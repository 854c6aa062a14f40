"""Accuracy of the recursion against the float64 oracle on a few lattice kinds, for the library FTR_LIB_PATH points at
(study builds: make -C tf-fast-rnnt_amd/csrc variant NAME=... DEFS=...).  python scripts/frames_study.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tf-fast-rnnt_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import rnnt_oracle as O
from helpers import max_rel
from tf_fast_rnnt.mutual_information import mi_forward_backward

def lattice(kind, B, S, T, seed=5):
    rng = np.random.default_rng(seed)
    if kind == "sharp":
        px = (rng.standard_normal((B, S, T + 1)) - 10.0).astype(np.float32)
        py = (rng.standard_normal((B, S + 1, T)) - 3.0).astype(np.float32)
        for b in range(B):
            ts = np.sort(np.clip(np.round((np.arange(S) + 0.5) * T / S).astype(int) + rng.integers(-3, 4, S), 0, T - 1))
            prev = 0
            for s_ in range(S + 1):
                end = ts[s_] if s_ < S else T
                py[b, s_, prev:end] = -0.05
                if s_ < S: px[b, s_, end] = -0.1
                prev = end
    elif kind == "blank_heavy":
        px = (0.5 * rng.standard_normal((B, S, T + 1)) - 8.0).astype(np.float32)
        py = (0.05 * rng.standard_normal((B, S + 1, T)) - 0.1).astype(np.float32)
    elif kind == "tilted":
        px = (rng.standard_normal((B, S, T + 1)) - 2.0).astype(np.float32)
        py = (rng.standard_normal((B, S + 1, T)) - 9.0).astype(np.float32)
    else:
        px = (rng.standard_normal((B, S, T + 1)) - 6.0).astype(np.float32)
        py = (rng.standard_normal((B, S + 1, T)) - 6.0).astype(np.float32)
    px[:, :, T] = -np.inf
    bd = np.zeros((B, 4), np.int32); bd[:, 2] = S; bd[:, 3] = T
    return px, py, bd

O.build()
dev = torch.device("cuda:0")
print("library:", os.environ.get("FTR_LIB_PATH", "default"))
for kind, B, S, T in (("iid", 2, 200, 1000), ("sharp", 2, 200, 1000), ("blank_heavy", 2, 200, 1000), ("tilted", 2, 200, 1000), ("iid", 1, 400, 3000), ("sharp", 1, 400, 3000), ("sharp", 2, 60, 1000)):
    px, py, bd = lattice(kind, B, S, T)
    a64, (gx64, gy64) = O.mutual_information_recursion(px, py, bd, True, np.float64)
    t = lambda a: torch.from_numpy(a).to(dev)
    ans, gx, gy = mi_forward_backward(t(px), t(py), t(bd), True)
    torch.cuda.synchronize()
    ans = ans.cpu().numpy().astype(np.float64)
    print(f"{kind:12s} S={S:4d} T={T:5d}  ans abs err {np.abs(ans - a64).max():.3e} rel {np.abs((ans - a64) / a64).max():.2e}   px_grad {max_rel(gx.cpu().numpy(), gx64):.2e}  py_grad {max_rel(gy.cpu().numpy(), gy64):.2e}", flush=True)
