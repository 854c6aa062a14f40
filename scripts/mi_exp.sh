# slot-time experiments: which wave bounds the bidirectional forward kernel (results are wrong by construction)
mkdir -p gpurun_out
for v in "" NODRAIN NOCOMPUTE NOPOLL; do
  if [ -z "$v" ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$PWD/gpurun_exp/libftr_$v.so; fi
  echo "== variant ${v:-product}"
  timeout -k 10 120 python scripts/mi_bench.py 32 200 1000 2>&1 | grep "warm" | head -1
  timeout -k 10 120 python scripts/mi_bench.py 32 63 1000 2>&1 | grep "warm" | head -1
done > gpurun_out/mi_exp.log 2>&1
cat gpurun_out/mi_exp.log
