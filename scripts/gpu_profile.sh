# rocprofv3 evidence for profiles/: bench lines of every BASELINE config, kernel stats of bench.py at c2 / c3 / c4 / c5, PMC
# FETCH_SIZE / WRITE_SIZE passes at c3 (separate runs, no trace domains next to --pmc).  Usage: bash scripts/gpu_profile.sh
mkdir -p gpurun_out
R=$PWD
for c in c3 c2 c4 c5; do
  st=20; [ $c = c5 ] && st=8; [ $c = c4 ] && st=10
  # also runs the one-off library-GEMM kernel search, outside the profiler (hundreds of candidate kernels); the profiled
  # runs below only apply its choices
  timeout -k 10 400 python bench.py --config $c --steps $st --warmup 4 --gemm-choices $R/gpurun_out/gemm_choices_$c.json $([ $c = c3 ] || echo --no-cpu-baseline) > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "bench $c exit $?"
done
cd /tmp && export TMPDIR=/tmp
for c in c3 c2 c4 c5; do
  st=10; [ $c = c5 ] && st=4; [ $c = c4 ] && st=5
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$c -- python3 $R/bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-graph --no-dense --gemm-choices $R/gpurun_out/gemm_choices_$c.json --no-gemm-search > $R/gpurun_out/prof_bench_$c.json 2> $R/gpurun_out/prof_$c.err; echo "prof $c exit $?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-dense --gemm-choices $R/gpurun_out/gemm_choices_c3.json --no-gemm-search > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err; echo "pmc fetch exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-dense --gemm-choices $R/gpurun_out/gemm_choices_c3.json --no-gemm-search > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err; echo "pmc write exit $?"
cd $R
# keep the merge-back small: the per-launch traces are large, the summaries are what profiles/ needs
python scripts/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic_c3.json > gpurun_out/pmc_summary.log 2>&1
for c in c3 c2 c4 c5; do f=$(ls gpurun_out/prof_$c/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_$c.csv; done
find gpurun_out -name "*_kernel_trace.csv" -delete
find gpurun_out -name "*_counter_collection.csv" -delete
du -sh gpurun_out
