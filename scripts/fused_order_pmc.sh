# HBM fetch traffic of the fused forward at c5 with the symbol tiles of a frame block side by side on one XCD (product) and in
# the previous order (study build forderold): one rocprofv3 --pmc FETCH_SIZE pass each (no other trace domain next to --pmc)
R=$PWD
B=$R/tf-fast-rnnt_amd/csrc/_build
cd /tmp && export TMPDIR=/tmp
for v in product forderold; do
  if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$B/libftr_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_forder_$v -- python3 $R/scripts/fused_bench.py 8 8000 1000 512 > $R/gpurun_out/pmc_forder_$v.log 2>&1
  echo "$v exit $?"
  python3 - "$R/gpurun_out/pmc_forder_$v" "$v" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "simple_fused_fwd" in k: print(sys.argv[2], "simple_fused_fwd_kernel: fetched per launch %.1f MB (2 x FETCH_SIZE KiB, %d launches)" % (2 * 1024 * sum(v) / len(v) / 1e6, len(v)))
PY
done
find $R/gpurun_out -name "*_counter_collection.csv" -delete; find $R/gpurun_out -name "*_kernel_trace.csv" -delete
