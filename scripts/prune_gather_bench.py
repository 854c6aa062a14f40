"""Times do_rnnt_pruning forward (ftr_do_pruning_f32) alone.  python scripts/prune_gather_bench.py [B T S C r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import tf_fast_rnnt as ft
B, T, S, C, r = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (32, 1000, 200, 500, 5)))
dev = torch.device("cuda:0"); g = torch.Generator(device="cpu").manual_seed(0)
am = torch.randn(B, T, C, generator=g).to(dev); lm = torch.randn(B, S + 1, C, generator=g).to(dev)
s0 = torch.clamp((torch.arange(T, dtype=torch.float32) * (S + 1 - r) / max(T - 1, 1)).round().long(), 0, max(S + 1 - r, 0))
ranges = (s0[None, :, None] + torch.arange(r)[None, None, :]).expand(B, T, r).contiguous().int().to(dev)
for _ in range(3): ft.do_rnnt_pruning(am, lm, ranges)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ft.do_rnnt_pruning(am, lm, ranges)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 20; mb = (B * T * r * C + B * (S + 1) * C + B * T * r) * 4 / 1e6   # the gather; am_pruned is a broadcast view
print(f"[{os.path.basename(os.environ.get('FTR_LIB_PATH', 'product'))}] do_rnnt_pruning B={B} T={T} S={S} C={C} r={r}: {us:.1f} us ({mb / us:.2f} TB/s algorithmic)")
