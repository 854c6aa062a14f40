# rocprofv3 evidence for profiles/: kernel stats of bench.py at c3 and c4, PMC FETCH_SIZE / WRITE_SIZE passes at c3 (separate runs)
mkdir -p gpurun_out
R=$PWD
# the library-GEMM kernel search runs once, outside the profiler (it launches hundreds of candidate kernels); the profiled
# runs only apply its choices
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --gemm-choices $R/gpurun_out/gemm_choices_c3.csv > /dev/null 2> gpurun_out/tune_c3.err; echo "tune c3 exit $?"
timeout -k 10 300 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-graph --gemm-choices $R/gpurun_out/gemm_choices_c4.csv > /dev/null 2> gpurun_out/tune_c4.err; echo "tune c4 exit $?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --gemm-choices $R/gpurun_out/gemm_choices_c3.csv --no-gemm-search > $R/gpurun_out/prof_bench_c3.json 2> $R/gpurun_out/prof_c3.err; echo "prof c3 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-graph --gemm-choices $R/gpurun_out/gemm_choices_c4.csv --no-gemm-search > $R/gpurun_out/prof_bench_c4.json 2> $R/gpurun_out/prof_c4.err; echo "prof c4 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --gemm-choices $R/gpurun_out/gemm_choices_c3.csv --no-gemm-search > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err; echo "pmc fetch exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --gemm-choices $R/gpurun_out/gemm_choices_c3.csv --no-gemm-search > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err; echo "pmc write exit $?"
cd $R
# keep the merge-back small: the per-launch traces are large, the summaries are what profiles/ needs
python scripts/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic_c3.json > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*_kernel_trace.csv" -delete
find gpurun_out -name "*_counter_collection.csv" -delete
du -sh gpurun_out
