mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest.log
grep -E "FAILED|passed|failed" gpurun_out/pytest.log | tail -15
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; echo "bench exit $rc" >> gpurun_out/bench.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench.json'))
print("graph", d.get("graph_replay")); print("ms/step", d["ms_per_step"], "value", d["value"], "native_us", d["native_us_per_step"], "peak MB", d["peak_mem_mb"], "roofline", d["roofline"])
for k,v in d["kernels"].items(): print(f"  {k:40s} {v['avg_us']:8.1f} us x{v['calls_per_step']}  {v['GBps']} GB/s")
PY
timeout -k 10 300 python bench.py --config c4 --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; rc=$?; echo "bench c4 exit $rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/bench_c4.json'))
    print("c4 ms/step", d["ms_per_step"], "value", d["value"], "native_us", d["native_us_per_step"], "peak MB", d["peak_mem_mb"])
    for k,v in d["kernels"].items(): print(f"  {k:40s} {v['avg_us']:8.1f} us x{v['calls_per_step']}  {v['GBps']} GB/s")
except Exception as e: print("c4 parse failed", e)
PY
if [ -z "$FTR_PROFILE" ]; then exit 0; fi
rm -rf gpurun_out/prof_r1 gpurun_out/pmc_fetch gpurun_out/pmc_write
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err; echo "prof exit $?" >> $R/gpurun_out/prof.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
rm -rf $R/gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_bench_c4.json 2> $R/gpurun_out/prof_c4.err; echo "prof c4 exit $?" >> $R/gpurun_out/prof_c4.err
cd $R; tail -2 gpurun_out/prof.err; ls gpurun_out/pmc_fetch/*/ | head -3
