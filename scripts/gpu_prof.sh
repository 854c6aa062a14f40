# kernel-trace profile of the bench step (no PMC): per-kernel average durations -> gpurun_out/prof_r1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_r1
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err; echo "prof exit $?"
cd $R; python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_r1/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:34]:
    n = r['Name'].replace('void ', '').replace('ftr::(anonymous namespace)::', '')[:70]
    print(f"{n:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
