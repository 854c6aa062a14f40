"""Times the fused builder forward (ftr_simple_logprobs_fused_fwd_f32) against library GEMM + epilogue on one shape.
python scripts/fused_bench.py [B T S C]"""
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
am = torch.randn(B, T, C, generator=g).to(dev); lm = torch.randn(B, S + 1, C, generator=g).to(dev)
sym = torch.randint(0, C - 1, (B, S), generator=g, dtype=torch.int32).to(dev)
amp = torch.empty_like(am); lmp = torch.empty_like(lm)
amx = torch.empty(B, T, device=dev); lmx = torch.empty(B, S + 1, device=dev)
px = torch.empty(B, S, T + 1, device=dev); py = torch.empty(B, S + 1, T, device=dev); prod = torch.empty(B, S + 1, T, device=dev)
px2 = torch.empty_like(px); py2 = torch.empty_like(py)
st = torch.cuda.current_stream().cuda_stream
_lib.call("ftr_rowmax_exp_f32", _ptr(am), _ptr(amp), _ptr(amx), B * T, C, st)
_lib.call("ftr_rowmax_exp_f32", _ptr(lm), _ptr(lmp), _ptr(lmx), B * (S + 1), C, st)
def fused(): _lib.call("ftr_simple_logprobs_fused_fwd_f32", _ptr(am), _ptr(lm), _ptr(sym), _ptr(amp), _ptr(lmp), _ptr(amx), _ptr(lmx), None, C - 1, 0.0, _ptr(px), _ptr(py), _ptr(prod), B, T, S, C, 0, st)
def library():
    p = torch.bmm(lmp, amp.transpose(1, 2))
    _lib.call("ftr_simple_logprobs_fwd_f32", _ptr(am), _ptr(lm), _ptr(sym), _ptr(p), _ptr(amx), _ptr(lmx), None, C - 1, 0.0, _ptr(px2), _ptr(py2), B, T, S, C, 0, st)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
tf, tl = timeit(fused), timeit(library)
fl = 2.0 * B * (S + 1) * T * C
print(f"B={B} T={T} S={S} C={C} [{os.environ.get('FTR_LIB_PATH', 'product')}]: fused {tf:.1f} us ({fl / tf / 1e6:.1f} TFLOP/s)   library GEMM + epilogue {tl:.1f} us   max |dpx| {(px[:, :, :T] - px2[:, :, :T]).abs().max().item():.2e} max |dpy| {(py - py2).abs().max().item():.2e}")
