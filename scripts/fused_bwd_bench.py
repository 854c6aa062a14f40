"""Times the backward of the simple builder on one shape: the fused d am kernel against W kernel + library GEMM + epilogue
kernel, and the whole backward (both routes).  python scripts/fused_bwd_bench.py [B T S C]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import tf_fast_rnnt as ft
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr
B, T, S, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 1000, 200, 500)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
amp = torch.rand(B, T, C, generator=g).to(dev); lmp = torch.rand(B, S + 1, C, generator=g).to(dev)
sym = torch.randint(0, C - 1, (B, S), generator=g, dtype=torch.int32).to(dev)
gx = torch.rand(B, S, T + 1, generator=g).to(dev); gy = torch.rand(B, S + 1, T, generator=g).to(dev)
prod = torch.bmm(lmp, amp.transpose(1, 2))
W = torch.empty_like(prod); rsx = torch.empty(B, S + 1, device=dev); rsy = torch.empty(B, S + 1, device=dev)
d_am = torch.empty_like(amp); d_am2 = torch.empty_like(amp); d_lm = torch.empty_like(lmp)
st = torch.cuda.current_stream().cuda_stream
def fused_am(): _lib.call("ftr_simple_logprobs_fused_bwd_am_f32", _ptr(gx), _ptr(gy), None, 0, 1.0, _ptr(prod), _ptr(lmp), _ptr(amp), _ptr(sym), None, C - 1, _ptr(d_am), B, T, S, C, 0, st)
def wk(): _lib.call("ftr_simple_logprobs_bwd_w_f32", _ptr(gx), _ptr(gy), _ptr(prod), None, _ptr(W), _ptr(rsx), _ptr(rsy), B, T, S, 0, st)
def lib_am():
    damp = torch.bmm(W.transpose(1, 2), lmp)
    _lib.call("ftr_simple_logprobs_bwd_am_f32", _ptr(gx), _ptr(gy), _ptr(damp), _ptr(amp), _ptr(sym), None, C - 1, _ptr(d_am2), B, T, S, C, 0, st)
def lib_lm():
    dlmp = torch.bmm(W, amp)
    _lib.call("ftr_simple_logprobs_bwd_lm_f32", _ptr(dlmp), _ptr(lmp), _ptr(sym), _ptr(rsx), _ptr(rsy), C - 1, _ptr(d_lm), B, S, C, st)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
wk()
fl = 2.0 * B * (S + 1) * T * C
tf = timeit(fused_am) if _lib.lib().ftr_simple_logprobs_fused_bwd_supported(T, C) else float("nan")
tw, ta, tl = timeit(wk), timeit(lib_am), timeit(lib_lm)
lib_am()
err = (d_am - d_am2).abs().max().item() / d_am2.abs().max().item() if tf == tf else float("nan")
print(f"B={B} T={T} S={S} C={C}: fused d_am {tf:.1f} us ({fl / tf / 1e6:.1f} TFLOP/s) | library: W {tw:.1f} + GEMM+d_am {ta:.1f} + GEMM+d_lm {tl:.1f} us | max rel diff d_am {err:.1e}")
