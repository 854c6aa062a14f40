mkdir -p gpurun_out
for v in "$@"; do
  export FTR_LIB_PATH=$PWD/gpurun_exp/libftr_$v.so
  for shape in "32 200 1000" "32 63 4000"; do
    timeout -k 10 100 python scripts/mi_stamps.py $shape 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/mi_stamps.log 2>&1
cat gpurun_out/mi_stamps.log
