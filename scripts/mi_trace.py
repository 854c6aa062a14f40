"""Kernel timeline of the recursion launches (diagnostic -DFTR_TRACE builds, FTR_LIB_PATH): kernel span on the device,
life of one traced workgroup and its per-slot times.  python scripts/mi_trace.py B S T"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mi_bench import run
from tf_fast_rnnt import _lib
B, S, T = (int(v) for v in sys.argv[1:4])
L = _lib.lib()
buf = (ctypes.c_ulonglong * 1024)()
f, b, _ = run(B, S, T, iters=3)
L.ftr_debug_trace(buf, 0)          # re-arm
f, b, _ = run(B, S, T, iters=1, warm=0)
L.ftr_debug_trace(buf, 1024)
v = list(buf)
t0 = v[0]
us = lambda x: (x - t0) / 100.0
n = v[4]
print(f"B={B} S={S} T={T}: events fwd {f:.1f} us bwd {b:.1f} us [{os.environ.get('FTR_LIB_PATH', 'product')}]")
print(f"  kernel span (first workgroup start -> last workgroup end): {us(v[1]):.2f} us; traced workgroup alive {us(v[2]):.2f} .. {us(v[3]):.2f} us, {n} slots")
ts = [us(x) for x in v[16:16 + n]]
print("  slot end times (us):", " ".join(f"{t:.2f}" for t in ts))
if n > 1:
    d = [ts[i + 1] - ts[i] for i in range(n - 1)]
    print("  slot durations (us):", " ".join(f"{t:.2f}" for t in d))
