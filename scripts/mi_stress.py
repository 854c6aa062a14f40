"""Determinism stress of the recursion kernels: many launches on the same inputs must give bit-identical results
(band hand-off timing varies from launch to launch; a race would show up as a mismatch or a NaN)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import tf_fast_rnnt as ft
from tf_fast_rnnt.mutual_information import mi_forward_backward
dev = torch.device("cuda:0")
bad = 0
for (B, S, T, mod, iters) in [(32, 200, 1000, False, 300), (32, 200, 1000, True, 150), (8, 1000, 8000, False, 40), (64, 130, 700, False, 150), (5, 300, 40, False, 200)]:
    g = torch.Generator(device="cpu").manual_seed(S + T)
    T1 = T if mod else T + 1
    px = (torch.randn((B, S, T1), generator=g) - 6.0).to(dev); py = (torch.randn((B, S + 1, T), generator=g) - 6.0).to(dev)
    bd = torch.zeros((B, 4), dtype=torch.int32); bd[:, 2] = S; bd[:, 3] = T
    bd[1, 2] = S // 2; bd[1, 3] = T // 2 + 1; bd = bd.to(dev)
    ref = None
    for i in range(iters):
        ans, gx, gy = mi_forward_backward(px, py, bd, True)
        if ref is None:
            ref = (ans.clone(), gx.clone(), gy.clone())
        elif not (torch.equal(ans, ref[0]) and torch.equal(gx, ref[1]) and torch.equal(gy, ref[2])):
            bad += 1
    torch.cuda.synchronize()
    print(f"B={B} S={S} T={T} modified={mod}: {iters} launches, finite={bool(torch.isfinite(ref[0]).all())}, mismatches so far {bad}", flush=True)
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
