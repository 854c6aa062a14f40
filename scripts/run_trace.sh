mkdir -p gpurun_out
for v in "$@"; do
  export FTR_LIB_PATH=$PWD/gpurun_exp/libftr_$v.so
  timeout -k 10 100 python scripts/mi_trace.py 32 200 1000 2>&1 | grep -v amdgpu.ids
done > gpurun_out/mi_trace.log 2>&1
cut -c1-330 gpurun_out/mi_trace.log | grep -v "slot end"
