"""Times ftr_prune_ranges_i32 alone (HIP events).  python scripts/prune_bench.py [B S T r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import tf_fast_rnnt as ft
B, S, T, r = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 200, 1000, 5)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
gx = torch.rand(B, S, T + 1, generator=g).to(dev); gy = torch.rand(B, S + 1, T, generator=g).to(dev)
bd = torch.tensor([[0, 0, S, T]] * B, dtype=torch.int32, device=dev)
for _ in range(3): ft.get_rnnt_prune_ranges(gx, gy, bd, r)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ft.get_rnnt_prune_ranges(gx, gy, bd, r)
e1.record(); torch.cuda.synchronize()
print(f"B={B} S={S} T={T} r={r}: get_rnnt_prune_ranges {e0.elapsed_time(e1) * 1000 / 20:.1f} us per call")
