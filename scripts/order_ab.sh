# A/B of study builds inside the step: product against _build/libftr_<name>.so for every name in VARIANTS
# (make -C tf-fast-rnnt_amd/csrc variant NAME=<name> SRC=... DEFS=...); two interleaved rounds; CALLS = native calls to print
for cfg in ${CFGS:-c3}; do
for round in 1 2; do
for v in product ${VARIANTS}; do
  if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$PWD/tf-fast-rnnt_amd/csrc/_build/libftr_$v.so; fi
  python bench.py --config $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-dense --no-graph > gpurun_out/b10.json 2>gpurun_out/b10.err
  CALLS="${CALLS:-ftr_do_pruning_f32 ftr_pruned_band_fwd_f32 ftr_pruned_band_bwd_scaled_f32 ftr_do_pruning_bwd_ws_f32}" python - <<PY
import json, os
d=json.load(open("gpurun_out/b10.json")); k=d["kernels"]
print("$cfg $v", d["ms_per_step"], {n[4:] if n.startswith("ftr_") else n: k[n]["avg_us"] for n in os.environ["CALLS"].split() if n in k})
PY
done; done; done
