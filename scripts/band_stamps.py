"""Phase times of the band-native recursion kernel (utterance 0): needs a -DFTR_BAND_STAMPS build (FTR_LIB_PATH).
python scripts/band_stamps.py [B T S r modified]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import tf_fast_rnnt as ft
from tf_fast_rnnt import _lib
B, T, S, r, mod = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (32, 1000, 200, 5, 0)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
# a plausible monotone band: s0[t] follows the diagonal
s0 = torch.clamp((torch.arange(T, dtype=torch.float32) * (S + 1 - r) / max(T - 1, 1)).round().long(), 0, max(S + 1 - r, 0))
ranges = (s0[None, :, None] + torch.arange(r)[None, None, :]).expand(B, T, r).contiguous().int().to(dev)
pxb = (-torch.rand(B, T, r, generator=g) * 3 - 0.1).to(dev)
pyb = (-torch.rand(B, T, r, generator=g) * 3 - 0.1).to(dev)
bnd = torch.tensor([[0, 0, S, T]] * B, dtype=torch.int32, device=dev)
ans = torch.empty(B, device=dev); gx = torch.empty_like(pxb); gy = torch.empty_like(pxb)
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.call("ftr_mutual_information_band_f32", pxb.data_ptr(), pyb.data_ptr(), ranges.data_ptr(), bnd.data_ptr(), ans.data_ptr(), gx.data_ptr(), gy.data_ptr(), B, T, S, r, mod, st)
for _ in range(5): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call()
e1.record(); torch.cuda.synchronize()
print(f"B={B} T={T} S={S} r={r} mod={mod}: {e0.elapsed_time(e1) * 1000 / 20:.1f} us per launch (events, back to back); ans[0]={ans[0].item():.3f}")
L = _lib.lib()
if hasattr(L, "ftr_debug_band_stamps"):
    buf = (ctypes.c_ulonglong * 16)()
    L.ftr_debug_band_stamps(buf)
    v = list(buf); names = ["start", "lo+fill", "stage X", "scatter X", "stage Y", "scatter Y", "forward", "cut", "flow", "store"]
    print("  " + " | ".join(f"{names[k]} {(v[k] - v[k - 1]) / 100.0:.2f}" for k in range(1, 10)) + f" | total {(v[9] - v[0]) / 100.0:.2f} us")
