# fused forward builder against the tuned library GEMM + epilogue, per BASELINE config.  Usage: bash scripts/builder_ab.sh
mkdir -p gpurun_out
for c in c3 c2 c4 c5; do
  st=16; [ $c = c5 ] && st=8; [ $c = c4 ] && st=8
  for mode in fused library fused library; do
    FTR_BUILDER_GEMM=$mode timeout -k 10 300 python bench.py --config $c --steps $st --warmup 4 --no-cpu-baseline --no-dense --no-graph > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "bench failed"; tail -5 gpurun_out/ab.err; exit 1; }
    python - "$c" "$mode" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json")); k = d["kernels"]
pick = {n: round(v["avg_us"], 1) for n, v in k.items() if "logprobs" in n and "pruned" not in n or "gemm" in n}
print(sys.argv[1], sys.argv[2], d["ms_per_step"], pick)
PY
  done
done
