# slot-time experiments: which wave bounds the bidirectional kernels (results of the variants are wrong by construction)
# usage: bash scripts/mi_exp.sh "SHAPE;SHAPE..." VARIANT...   (libraries gpurun_exp/libftr_<VARIANT>.so, "product" = the in-tree library)
mkdir -p gpurun_out
shapes=$1; shift
for v in "$@"; do
  if [ "$v" = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$PWD/gpurun_exp/libftr_$v.so; fi
  echo "== variant $v"
  IFS=';' read -ra SH <<< "$shapes"
  for shape in "${SH[@]}"; do
    timeout -k 10 120 python scripts/mi_bench.py $shape 2>&1 | grep "warm" | head -1
  done
done > gpurun_out/mi_exp.log 2>&1
cat gpurun_out/mi_exp.log
