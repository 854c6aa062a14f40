"""Accuracy of the recursion against the float64 oracle: native (default family), the one-ended chain, the plain
(reference arithmetic) family and the float32 oracle, normwise relative error of ans and of the occupancies."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import tf_fast_rnnt as ft
import rnnt_oracle as O
from helpers import max_rel
from tf_fast_rnnt.mutual_information import mi_forward_backward
O.build()
dev = torch.device("cuda:0")
for (B, S, T) in [(2, 50, 200), (2, 100, 512), (2, 200, 1000), (1, 400, 3000)]:
    rng = np.random.default_rng(S + T)
    px = (rng.standard_normal((B, S, T + 1)) - 6.0).astype(np.float32); px[:, :, T] = -np.inf
    py = (rng.standard_normal((B, S + 1, T)) - 6.0).astype(np.float32)
    bd = np.zeros((B, 4), np.int32); bd[:, 2] = S; bd[:, 3] = T
    a64, p64 = O.mi_forward(px, py, bd, dtype=np.float64); gx64, gy64, _ = O.mi_backward(px, py, bd, p64, dtype=np.float64)
    a32, p32 = O.mi_forward(px, py, bd); gx32, gy32, _ = O.mi_backward(px, py, bd, p32)
    line = f"B={B} S={S} T={T}:  oracle f32 vs f64  ans {np.max(np.abs(a32 - a64) / np.abs(a64)):.1e} grads {max(max_rel(gx32, gx64), max_rel(gy32, gy64)):.1e}"
    for name, impl in (("bidir", 0), ("chain", 4), ("plain", 1)):
        if impl == 1 and S + 1 > 1024: continue
        prev = ft._lib.lib().ftr_set_mi_impl(impl)
        t = lambda a: torch.from_numpy(a).to(dev)
        ans, gx, gy = mi_forward_backward(t(px), t(py), t(bd), True)
        torch.cuda.synchronize(); ft._lib.lib().ftr_set_mi_impl(prev)
        line += f" | {name} ans {np.max(np.abs(ans.cpu().numpy() - a64) / np.abs(a64)):.1e} grads {max(max_rel(gx.cpu().numpy(), gx64), max_rel(gy.cpu().numpy(), gy64)):.1e}"
    print(line, flush=True)
