"""How do the synthetic ranges of bench.py look to do_pruning_bwd (chunk spans, partial rows per lattice row), and what do
its two kernels cost separately?  Usage: python scripts/prune_bwd_study.py [c3|c4|c5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import numpy as np
import torch
import bench
import tf_fast_rnnt as ft

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
B, T, S, C, r = bench.CONFIGS[cfg]
dev = torch.device("cuda:0")
inp = bench.make_inputs(B, T, S, C, 1234, dev)
out = bench.pruned_step(inp, r, keep=True, first_pass="smoothed" if cfg == "c4" else "simple")
rg = out["ranges"].cpu().numpy()          # [B,T,r]
base = rg[:, :, 0]
d = np.diff(base, axis=1)
print(f"{cfg}: ranges[b,t,0] steps: min {d.min()} max {d.max()} mean {d.mean():.4f}; flat {np.mean(d == 0):.3f}")
for TCH in (16, 32, 64):
    nch = (T + TCH - 1) // TCH
    spans = []
    hits = np.zeros((B, S + 1), np.int32)
    for b in range(B):
        for ch in range(nch):
            t0, t1 = ch * TCH, min(T, ch * TCH + TCH)
            lo, hi = base[b, t0], base[b, t1 - 1] + r - 1
            spans.append(hi - lo + 1)
            hits[b, lo:hi + 1] += 1
    spans = np.array(spans)
    print(f"  TCH={TCH}: span mean {spans.mean():.1f} max {spans.max()} p99 {np.percentile(spans, 99):.0f}; "
          f"partial rows {spans.sum()} ({spans.sum() * C * 4 / 1e6:.1f} MB); hits per (b,s): mean {hits.mean():.2f} max {hits.max()}")
g = torch.randn((B, T, r, C), device=dev)
g2 = g.clone() if os.environ.get("STUDY_TWO_TENSORS") else g
d_am = torch.empty((B, T, C), device=dev); d_lm = torch.empty((B, S + 1, C), device=dev)
L = ft._lib
ws_bytes = L.lib().ftr_do_pruning_bwd_workspace_bytes(B, T, S + 1, C, r)
ws = torch.empty(ws_bytes // 4 + 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    L.call("ftr_do_pruning_bwd_ws_f32", g2.data_ptr(), g.data_ptr(), out["ranges"].data_ptr(), d_am.data_ptr(), d_lm.data_ptr(),
           B, T, S + 1, C, r, ws.data_ptr(), ws_bytes, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"  do_pruning_bwd (both kernels, warm, isolated): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
# reference result for a correctness spot check
ref_am = g.sum(2)
ref_lm = torch.zeros_like(d_lm)
idx = out["ranges"].long().reshape(B, T * r)
for b in range(B):
    ref_lm[b].index_add_(0, idx[b], g[b].reshape(T * r, C))
print("  d_am err", float((d_am - ref_am).abs().max()), " d_lm err", float((d_lm - ref_lm).abs().max()))
