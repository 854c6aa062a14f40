# A/B of a study build inside the step: product against _build/libftr_$VARIANT.so (make -C tf-fast-rnnt_amd/csrc variant ...),
# optionally with FTR_PRUNE_SEG set for both; prints ms per step and the calls whose order of walking memory was studied
for cfg in ${CFGS:-c3}; do
for v in product ${VARIANT:-product} product ${VARIANT:-product}; do
  if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$PWD/tf-fast-rnnt_amd/csrc/_build/libftr_$v.so; fi
  python bench.py --config $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-dense --no-graph > gpurun_out/b10.json 2>gpurun_out/b10.err
  python - <<PY
import json
d=json.load(open("gpurun_out/b10.json")); k=d["kernels"]
print("$cfg $v seg=${FTR_PRUNE_SEG:-default}", d["ms_per_step"], {n[4:]: k[n]["avg_us"] for n in ("ftr_do_pruning_f32","ftr_pruned_band_fwd_f32","ftr_pruned_band_bwd_scaled_f32","ftr_do_pruning_bwd_ws_f32")})
PY
done; done
