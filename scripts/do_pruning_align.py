"""do_pruning_kernel's two speed modes (126 vs 166 us at c3): is it the relative placement of the two 320 MB output streams?
Times ftr_do_pruning_f32 with lm_pruned placed at am_pruned + N + delta for several deltas (one big allocation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
from tf_fast_rnnt import _lib
dev = torch.device("cuda:0")
B, T, S, C, r = 32, 1000, 200, 500, 5
g = torch.Generator(device="cpu").manual_seed(0)
am = torch.randn((B, T, C), generator=g).to(dev); lm = torch.randn((B, S + 1, C), generator=g).to(dev)
s0 = (torch.arange(T) * (S - r + 1) // T).to(torch.int32)
ranges = (s0[None, :, None] + torch.arange(r, dtype=torch.int32)[None, None, :]).expand(B, T, r).contiguous().to(dev)
N = B * T * r * C
big = torch.empty(2 * N + (64 << 20) // 4, dtype=torch.float32, device=dev)
base = big.data_ptr()
st = torch.cuda.current_stream().cuda_stream
def run(delta_bytes, iters=30):
    a_ptr = base
    l_ptr = base + 4 * N + delta_bytes
    for _ in range(3):
        _lib.call("ftr_do_pruning_f32", am.data_ptr(), lm.data_ptr(), ranges.data_ptr(), a_ptr, l_ptr, B, T, S + 1, C, r, st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.call("ftr_do_pruning_f32", am.data_ptr(), lm.data_ptr(), ranges.data_ptr(), a_ptr, l_ptr, B, T, S + 1, C, r, st)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
print(f"N = {4*N} bytes per output; base % 2MiB = {base % (2<<20)}")
for d in [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (2 << 20) - (4 * N) % (2 << 20), (2 << 20) - (4 * N) % (2 << 20) + 4096, 3 << 20, (1 << 20) + 256]:
    print(f"delta {d:9d} B  (lm_p - am_p) % 2MiB = {(4*N + d) % (2<<20):8d}: {run(d):7.1f} us", flush=True)
