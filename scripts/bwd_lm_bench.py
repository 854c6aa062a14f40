"""Times ftr_simple_logprobs_bwd_lm_f32 / ftr_smoothed_logprobs_bwd_lm_f32 alone and prints checksums of their outputs
(bit-equality across builds).  python scripts/bwd_lm_bench.py [B T S C]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr
B, T, S, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 1000, 200, 500)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
lmp = torch.rand(B, S + 1, C, generator=g).to(dev); dlmp = torch.randn(B, S + 1, C, generator=g).to(dev)
sym = torch.randint(0, C - 1, (B, S), generator=g, dtype=torch.int32).to(dev)
rsx = torch.randn(B, S + 1, generator=g).to(dev); rsy = torch.randn(B, S + 1, generator=g).to(dev)
arow = torch.randn(B, S + 1, generator=g).to(dev); invsum = torch.rand(B, S + 1, generator=g).to(dev); gu = torch.randn(C, generator=g).to(dev)
d1 = torch.empty_like(lmp); d2 = torch.empty_like(lmp)
st = torch.cuda.current_stream().cuda_stream
def simple(): _lib.call("ftr_simple_logprobs_bwd_lm_f32", _ptr(dlmp), _ptr(lmp), _ptr(sym), _ptr(rsx), _ptr(rsy), C - 1, _ptr(d1), B, S, C, st)
def smoothed(): _lib.call("ftr_smoothed_logprobs_bwd_lm_f32", _ptr(dlmp), _ptr(lmp), _ptr(sym), _ptr(rsx), _ptr(rsy), C - 1, 0.7, _ptr(arow), _ptr(invsum), _ptr(gu), _ptr(d2), B, S, C, st)
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / 20
sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]
t1, t2 = timeit(simple), timeit(smoothed)
print(f"B={B} S={S} C={C}: d_lm simple {t1:.1f} us sha {sha(d1)}   smoothed {t2:.1f} us sha {sha(d2)}")
