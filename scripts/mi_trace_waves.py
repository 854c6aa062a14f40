"""Which wave of the traced forward band reaches the barrier last, slot by slot (diagnostic -DFTR_TRACE=3 builds, FTR_LIB_PATH):
python scripts/mi_trace_waves.py B S T"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mi_bench import run
from tf_fast_rnnt import _lib
B, S, T = (int(v) for v in sys.argv[1:4])
L = _lib.lib(); buf = (ctypes.c_ulonglong * 1024)()
run(B, S, T, iters=3); L.ftr_debug_trace(buf, 0)
f, b, _ = run(B, S, T, iters=1, warm=0); L.ftr_debug_trace(buf, 1024)
v = list(buf); t0 = v[0]; n = min(int(v[4]), 240)
names = ["compute", "IO-in", "COMM", "IO-out"]
print(f"[{os.path.basename(os.environ.get('FTR_LIB_PATH', 'product'))}] fwd {f:.1f} us, span {(v[1] - t0) / 100:.1f} us, {n} slots; arrival at the barrier, us after the previous slot's last arrival:")
prev = (v[2] - t0) / 100.0
for i in range(min(n, int(os.environ.get("FTR_TRACE_N", "24")))):
    a = [(v[256 * w + 16 + i] - t0) / 100.0 for w in range(4)]
    last = max(a)
    print(f"  slot {i:3d} ends {last:7.2f}  (+{last - prev:5.2f})  " + "  ".join(f"{names[w]} {a[w] - prev:5.2f}" for w in range(4)) + f"   last: {names[a.index(last)]}")
    prev = last
