import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tf-fast-rnnt_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import tf_fast_rnnt as ft, rnnt_oracle as O
from tf_fast_rnnt import _lib
from helpers import synthetic
O.build()
np.set_printoptions(linewidth=250, precision=2, suppress=True)
dev = torch.device("cuda:0")
T, S, r, mod, B, C = 12, 6, 3, 0, 1, 12
d = synthetic(7 + S, B, T, S, C, ragged=False)
blank = d["termination_symbol"]
_, (gx, gy) = O.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], blank, d["boundary"], reduction="sum", calc_gradients=True)
ranges = O.get_rnnt_prune_ranges(gx, gy, d["boundary"], r)
am_p, lm_p = O.do_rnnt_pruning(d["am"], d["lm"], ranges)
logits = np.tanh(am_p + lm_p).astype(np.float32)
px, py = O.get_rnnt_logprobs_pruned(logits, d["symbols"], ranges, blank, d["boundary"])
ans, p = O.mi_forward(px, py, d["boundary"], dtype=np.float64)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tl, ts, tr, tb = t(logits), t(d["symbols"]), t(ranges), t(d["boundary"])
lse = torch.empty((B, T, r), device=dev); pxb = torch.empty_like(lse); pyb = torch.empty_like(lse)
gxb = torch.full((B, T, r), 777.0, device=dev); gyb = torch.full((B, T, r), 777.0, device=dev); a = torch.empty(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.call("ftr_pruned_band_fwd_f32", tl.data_ptr(), ts.data_ptr(), tr.data_ptr(), tb.data_ptr(), blank, 0.0, lse.data_ptr(), pxb.data_ptr(), pyb.data_ptr(), B, T, S, C, r, mod, st)
_lib.call("ftr_mutual_information_band_f32", pxb.data_ptr(), pyb.data_ptr(), tr.data_ptr(), tb.data_ptr(), a.data_ptr(), gxb.data_ptr(), gyb.data_ptr(), B, T, S, r, mod, st)
torch.cuda.synchronize()
s0 = ranges[0, :, 0]
print("s0", s0.tolist(), "ans", a.item(), "oracle", ans)
pb = np.stack([[p[0, s0[tt] + k, tt] if s0[tt] + k <= S else np.nan for k in range(r)] for tt in range(T)])
print("oracle alpha (band, rows t):\n", pb.T)
print("gpu chain A values:\n", gxb.cpu().numpy()[0].T)
print("gpu chain B values:\n", gyb.cpu().numpy()[0].T)
