"""Times ftr_simple_logprobs_bwd_am_f32 (the d am epilogue behind the library GEMM) alone on one shape and prints a checksum
of its output (bit-equality across study builds).  python scripts/bwd_am_bench.py [B T S C]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr
B, T, S, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 1000, 200, 500)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
amp = torch.rand(B, T, C, generator=g).to(dev); damp = torch.randn(B, T, C, generator=g).to(dev)
sym = torch.randint(0, C - 1, (B, S), generator=g, dtype=torch.int32).to(dev)
gx = torch.rand(B, S, T + 1, generator=g).to(dev); gy = torch.rand(B, S + 1, T, generator=g).to(dev)
d_am = torch.empty_like(amp)
st = torch.cuda.current_stream().cuda_stream
def run(): _lib.call("ftr_simple_logprobs_bwd_am_f32", _ptr(gx), _ptr(gy), _ptr(damp), _ptr(amp), _ptr(sym), None, C - 1, _ptr(d_am), B, T, S, C, 0, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 20
mb = 4.0 * (3 * B * T * C + B * S * (T + 1) + B * (S + 1) * T) / 1e6
print(f"B={B} T={T} S={S} C={C}: d_am epilogue {us:.1f} us ({mb / us / 1e6 * 1e6 / 1e6:.2f} TB/s of {mb:.0f} MB)  sha {hashlib.sha1(d_am.cpu().numpy().tobytes()).hexdigest()[:12]}")
