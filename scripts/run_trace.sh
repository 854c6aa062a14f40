mkdir -p gpurun_out
for v in TRF TRB TRF3 TRB3; do
  export FTR_LIB_PATH=$PWD/gpurun_exp/libftr_$v.so
  for shape in "32 200 1000" "32 63 1000"; do
    case "$v$shape" in TRF332\ 63*|TRB332\ 63*) continue;; esac
    timeout -k 10 100 python scripts/mi_trace.py $shape 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/mi_trace.log 2>&1
cat gpurun_out/mi_trace.log
