"""Bisects wavefront 'duo' vs 'mono' kernels: G workspace and gradients for one small lattice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import tf_fast_rnnt as ft
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr
from helpers import random_lattice
import rnnt_oracle as O
np.set_printoptions(precision=4, linewidth=200, suppress=True)
dev = torch.device("cuda:0")
L = _lib.lib()
B, S, T = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 4, 8)
px, py, bd = random_lattice(100 + S + T, B, S, T, modified=False, ragged=True)
tpx, tpy, tbd = (torch.from_numpy(a).to(dev) for a in (px, py, bd))
st = torch.cuda.current_stream().cuda_stream
def fwd(impl):
    L.ftr_set_mi_impl(impl)
    ws = torch.full((L.ftr_mutual_information_workspace_floats(B, S, T),), -7.0, device=dev); ans = torch.empty(B, device=dev)
    _lib.call("ftr_mutual_information_fwd_f32", _ptr(tpx), _ptr(tpy), _ptr(tbd), _ptr(ws), _ptr(ans), B, S, T, 0, st)
    torch.cuda.synchronize(); return ws, ans
def bwd(impl, ws):
    L.ftr_set_mi_impl(impl)
    gx = torch.full_like(tpx, -9.0); gy = torch.full_like(tpy, -9.0); ag = torch.ones(B, device=dev)
    _lib.call("ftr_mutual_information_bwd_f32", _ptr(tpx), _ptr(tpy), _ptr(tbd), _ptr(ws), None, _ptr(gx), _ptr(gy), _ptr(ag), 1, B, S, T, 0, st)
    torch.cuda.synchronize(); return gx.cpu().numpy(), gy.cpu().numpy(), ag.cpu().numpy()
ws_d, ans_d = fwd(0); ws_m, ans_m = fwd(2)  # 0 = default (chain), 2 = mono
print("boundary", bd.tolist()); print("ans duo", ans_d.tolist(), "mono", ans_m.tolist())
nL = B * (S + 1) * (T + 1)
Gd = ws_d.cpu().numpy()[:nL].reshape(B, S + 1, T + 1); Gm = ws_m.cpu().numpy()[:nL].reshape(B, S + 1, T + 1)
print("max |G duo - G mono|", np.abs(Gd - Gm).max())
if np.abs(Gd - Gm).max() > 1e-6:
    bad = np.argwhere(np.abs(Gd - Gm) > 1e-6); print("first differing cells", bad[:20].tolist())
    print("G duo b0\n", Gd[0]); print("G mono b0\n", Gm[0])
o_ans, o_p = O.mi_forward(px, py, bd); o_gx, o_gy, _ = O.mi_backward(px, py, bd, o_p)
for fi, fn in ((0, "duo"), (2, "mono")):
    for bi, bn in ((0, "duo"), (2, "mono")):
        gx, gy, ag = bwd(bi, ws_d if fi == 0 else ws_m)
        print(f"fwd {fn} + bwd {bn}: max|gx-oracle| {np.abs(gx - o_gx).max():.3g}  max|gy-oracle| {np.abs(gy - o_gy).max():.3g}  check {ag.tolist()}")
gx, gy, _ = bwd(0, ws_m)
if np.abs(gx - o_gx).max() > 1e-4:
    print("gx duo-bwd b0\n", gx[0]); print("gx oracle b0\n", o_gx[0]); print("gy duo-bwd b0\n", gy[0]); print("gy oracle b0\n", o_gy[0])
L.ftr_set_mi_impl(0)
