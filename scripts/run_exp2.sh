./scripts/micro/slotbench.bin > gpurun_out/slotbench.log 2>&1
for v in product FNONE FNOSL; do
  if [ "$v" = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$PWD/gpurun_exp/libftr_$v.so; fi
  echo "== variant $v"
  for shape in "32 63 1000" "32 63 4000" "32 63 8000" "8 63 8000" "128 63 1000"; do
    timeout -k 10 120 python scripts/mi_bench.py $shape 2>&1 | grep "warm" | head -1
  done
done > gpurun_out/mi_exp2.log 2>&1
cat gpurun_out/slotbench.log gpurun_out/mi_exp2.log
