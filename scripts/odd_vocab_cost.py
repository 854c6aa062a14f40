"""What a vocabulary size that is not a multiple of 4 costs: the simple loss forward + backward at C and C + 1 (the 16-byte
kernels -- fused builder, row kernels -- need C % 4 == 0; other sizes take the library-GEMM route and the scalar paths).
python scripts/odd_vocab_cost.py [B T S C]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch, bench
B, T, S, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 1000, 200, 500)))
dev = torch.device("cuda:0")
for c in (C, C + 1):
    inp = bench.make_inputs(B, T, S, c, 0, dev)
    for _ in range(4): bench.simple_step(inp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bench.simple_step(inp)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} T={T} S={S} C={c}: rnnt_loss_simple forward + backward {e0.elapsed_time(e1) * 50:.1f} us")
