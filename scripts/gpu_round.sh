mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest.log
grep -E "FAILED|passed|failed" gpurun_out/pytest.log | tail -15
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; echo "bench exit $rc" >> gpurun_out/bench.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err; echo "prof exit $?" >> $R/gpurun_out/prof.err
cd $R; cat gpurun_out/bench.json | cut -c1-1700; tail -2 gpurun_out/bench.err
# stamped diagnostic build (scratch copy only)
make -C tf-fast-rnnt_amd/csrc clean > /dev/null; make -C tf-fast-rnnt_amd/csrc -j8 STAMPS=1 > gpurun_out/stamp_build.log 2>&1
timeout -k 10 200 python scripts/mi_bench.py 32 200 1000 > gpurun_out/stamps_duo.log 2>&1; timeout -k 10 200 python scripts/mi_bench.py 32 63 1000 >> gpurun_out/stamps_duo.log 2>&1; grep -v amdgpu.ids gpurun_out/stamps_duo.log
