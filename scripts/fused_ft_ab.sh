# A/B of the fused forward's frame tile (FTR_FUSED_FT = 64 | 128): bench lines per config
for cfg in ${CFGS:-c4 c3 c5}; do
  for ft in 64 128; do
    FTR_FUSED_FT=$ft python bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-dense --no-graph > gpurun_out/ft_${cfg}_$ft.json 2> gpurun_out/ft_${cfg}_$ft.err
    python - <<PY
import json
d=json.load(open("gpurun_out/ft_${cfg}_$ft.json"))
k=d["kernels"]
name=[n for n in k if "fused_fwd" in n][0]
print("$cfg frames=$ft ms/step", d["ms_per_step"], name, k[name]["avg_us"], "loss", d["loss"])
PY
  done
done
