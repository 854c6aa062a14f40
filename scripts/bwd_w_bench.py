"""Times ftr_simple_logprobs_bwd_w_f32 alone and prints checksums of its outputs.  python scripts/bwd_w_bench.py [B T S C]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr
B, T, S, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 1000, 200, 500)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
gx = torch.rand(B, S, T + 1, generator=g).to(dev); gy = torch.rand(B, S + 1, T, generator=g).to(dev)
prod = torch.rand(B, S + 1, T, generator=g).to(dev) + 0.1
W = torch.empty_like(prod); rsx = torch.empty(B, S + 1, device=dev); rsy = torch.empty(B, S + 1, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(): _lib.call("ftr_simple_logprobs_bwd_w_f32", _ptr(gx), _ptr(gy), _ptr(prod), None, _ptr(W), _ptr(rsx), _ptr(rsy), B, T, S, 0, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10]
print(f"B={B} T={T} S={S}: W kernel {e0.elapsed_time(e1) * 50:.1f} us  sha {sha(W)} {sha(rsx)} {sha(rsy)}")
