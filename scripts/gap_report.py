"""Idle time between consecutive kernels of bench.py's timed steps, from a rocprofv3 --kernel-trace csv:
python scripts/gap_report.py <..._kernel_trace.csv> [steps]  -> the largest gaps by (previous kernel -> next kernel)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ftr::(anonymous namespace)::", "").replace("ftr::(anonymous namespace)::", "").split("(")[0][:48]
# the timed region: the last `steps` occurrences of the first kernel of a step
first = "rowmax_exp_kernel"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# two rowmax_exp launches per step (am, lm)
begin = starts[-2 * steps]
sel = rows[begin:]
gaps = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for a, b in zip(sel, sel[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    busy += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    key = (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
    gaps[key][0] += 1; gaps[key][1] += max(g, 0.0)
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
print(f"{steps} steps: span {span / steps:.1f} us/step, kernels busy {busy / steps:.1f} us/step, idle {(span - busy) / steps:.1f} us/step")
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {t / steps:7.1f} us/step  ({n / steps:.1f}x {t / n:5.1f} us)  {a} -> {b}")
