mkdir -p gpurun_out
timeout -k 10 120 python scripts/mi_debug.py > gpurun_out/mi_debug.log 2>&1; grep -v amdgpu.ids gpurun_out/mi_debug.log | head -12
timeout -k 10 120 python scripts/mi_debug.py 3 130 90 > gpurun_out/mi_debug2.log 2>&1; grep -E "max|fwd" gpurun_out/mi_debug2.log | head -8
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest.log
grep -E "FAILED|passed|failed" gpurun_out/pytest.log | tail -15
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python scripts/mi_bench.py > gpurun_out/mi_bench.log 2>&1; grep -v amdgpu.ids gpurun_out/mi_bench.log | grep warm
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; echo "bench exit $rc" >> gpurun_out/bench.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err; echo "prof exit $?" >> $R/gpurun_out/prof.err
cd $R; python - <<'PY'
import json
d = json.load(open('gpurun_out/bench.json'))
print("ms/step", d["ms_per_step"], "value", d["value"], "native_us", d["native_us_per_step"], "peak MB", d["peak_mem_mb"], "roofline", d["roofline"])
for k,v in d["kernels"].items(): print(f"  {k:40s} {v['avg_us']:8.1f} us x{v['calls_per_step']}  {v['GBps']} GB/s")
PY
tail -2 gpurun_out/bench.err
