"""bidirectional (impl 0) vs one-ended chain (impl 4) mutual-information kernels over batch sizes and band counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mi_bench import run
for (B, S, T) in [(8, 63, 1000), (32, 63, 1000), (64, 63, 1000), (128, 63, 1000), (8, 200, 1000), (16, 200, 1000), (32, 200, 1000), (64, 200, 1000), (8, 1000, 8000), (32, 300, 2000)]:
    r = {}
    for impl in (0, 4):
        r[impl] = run(B, S, T, impl=impl)
    print(f"B={B:4d} S={S:5d} T={T:5d}: bidir fwd {r[0][0]:7.1f} bwd {r[0][1]:7.1f} us | chain fwd {r[4][0]:7.1f} bwd {r[4][1]:7.1f} us", flush=True)
