"""Slot-time summary of one traced band of the recursion forward (diagnostic -DFTR_TRACE builds, FTR_LIB_PATH):
python scripts/mi_trace_summary.py B S T"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mi_bench import run
from tf_fast_rnnt import _lib
import numpy as np
B, S, T = (int(v) for v in sys.argv[1:4])
L = _lib.lib()
buf = (ctypes.c_ulonglong * 1024)()
run(B, S, T, iters=3)
L.ftr_debug_trace(buf, 0)
f, b, _ = run(B, S, T, iters=1, warm=0)
L.ftr_debug_trace(buf, 1024)
v = list(buf); t0 = v[0]; us = lambda x: (x - t0) / 100.0
n = int(v[4]); ts = np.array([us(x) for x in v[16:16 + n]]); d = np.diff(ts)
q = max(len(d) // 4, 1)
if os.environ.get("FTR_TRACE_SLOTS"):
    print("  slot end times (us):", " ".join(f"{t:.2f}" for t in ts))
print(f"[{os.path.basename(os.environ.get('FTR_LIB_PATH', 'product'))}] fwd {f:.1f} us; kernel span {us(v[1]):.1f}; band alive {us(v[2]):.1f} .. {us(v[3]):.1f} us, {n} slots; "
      f"slot us: first quarter {d[:q].mean():.3f}, middle {d[q:3*q].mean():.3f}, last quarter {d[3*q:].mean():.3f}; max {d.max():.2f}; slots > 1.2 us: {(d > 1.2).sum()}")
