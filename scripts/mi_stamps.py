"""Reads the per-wave slot stamps of the bidirectional forward kernel (diagnostic STAMPS builds, FTR_LIB_PATH)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mi_bench import run
from tf_fast_rnnt import _lib
B, S, T = (int(v) for v in sys.argv[1:4])
f, b, _ = run(B, S, T, iters=5)
buf = (ctypes.c_ulonglong * 16)()
_lib.lib().ftr_debug_stamps(buf)
v = list(buf)
names = ["compute", "IO-in", "COMM", "IO-out"]
print(f"B={B} S={S} T={T}: fwd {f:.1f} us bwd {b:.1f} us   [{os.environ.get('FTR_LIB_PATH', 'product')}]")
for i, n in enumerate(names):
    cnt = max(v[4 * i + 2], 1)
    print(f"   {n:8s} busy {v[4*i]/cnt:8.0f}  barrier-wait {v[4*i+1]/cnt:8.0f} ticks/slot  ({cnt} slots, {(v[4*i]+v[4*i+1])/cnt/2.4e3:.2f} us/slot at 2.4 GHz)")
