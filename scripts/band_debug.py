"""debug helper: the band kernel against the CPU oracle on single utterances of growing height."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tf-fast-rnnt_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import tf_fast_rnnt as ft, rnnt_oracle as O
from tf_fast_rnnt import _lib
from helpers import synthetic
O.build()
dev = torch.device("cuda:0")
for (T, S, r, mod) in [(20, 6, 3, 0), (20, 7, 4, 0), (20, 8, 4, 0), (20, 9, 4, 0), (40, 12, 4, 0), (40, 20, 5, 0), (20, 9, 4, 1), (40, 20, 5, 1)]:
    B, C = 1, 12
    d = synthetic(7 + S, B, T, S, C, ragged=False)
    blank = d["termination_symbol"]; rt = "modified" if mod else "regular"
    _, (gx, gy) = O.rnnt_loss_simple(d["lm"], d["am"], d["symbols"], blank, d["boundary"], rnnt_type=rt, reduction="sum", calc_gradients=True)
    ranges = O.get_rnnt_prune_ranges(gx, gy, d["boundary"], r)
    am_p, lm_p = O.do_rnnt_pruning(d["am"], d["lm"], ranges)
    logits = np.tanh(am_p + lm_p).astype(np.float32)
    px, py = O.get_rnnt_logprobs_pruned(logits, d["symbols"], ranges, blank, d["boundary"], rt)
    ans, (ogx, ogy) = O.mutual_information_recursion(px, py, d["boundary"], True, np.float64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tl, ts, tr, tb = t(logits), t(d["symbols"]), t(ranges), t(d["boundary"])
    lse = torch.empty((B, T, r), device=dev); pxb = torch.empty_like(lse); pyb = torch.empty_like(lse)
    gxb = torch.empty_like(lse); gyb = torch.empty_like(lse); a = torch.empty(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("ftr_pruned_band_fwd_f32", tl.data_ptr(), ts.data_ptr(), tr.data_ptr(), tb.data_ptr(), blank, 0.0, lse.data_ptr(), pxb.data_ptr(), pyb.data_ptr(), B, T, S, C, r, mod, st)
    _lib.call("ftr_mutual_information_band_f32", pxb.data_ptr(), pyb.data_ptr(), tr.data_ptr(), tb.data_ptr(), a.data_ptr(), gxb.data_ptr(), gyb.data_ptr(), B, T, S, r, mod, st)
    torch.cuda.synchronize()
    s0 = ranges[0, :, 0]
    T1 = px.shape[2]
    o_pxb = np.stack([[px[0, s0[tt] + k, tt] if s0[tt] + k < S else -np.inf for k in range(r)] for tt in range(T)])
    ogxb = np.stack([[ogx[0, s0[tt] + k, tt] if s0[tt] + k < S else 0 for k in range(r)] for tt in range(T)])
    ogyb = np.stack([[ogy[0, s0[tt] + k, tt] if s0[tt] + k <= S else 0 for k in range(r)] for tt in range(T)])
    e_in = np.nanmax(np.abs(np.where(np.isfinite(o_pxb), pxb.cpu().numpy()[0] - o_pxb, 0)))
    print(f"T={T} S={S} r={r} mod={mod}: ans gpu {a.item():.5f} oracle {ans[0]:.5f} | band px err {e_in:.2e} | gx err {np.abs(gxb.cpu().numpy()[0]-ogxb).max():.2e} gy err {np.abs(gyb.cpu().numpy()[0]-ogyb).max():.2e}", flush=True)
