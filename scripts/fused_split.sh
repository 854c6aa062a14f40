# contraction / epilogue split of the fused forward: study builds FTR_FUSED_EXP=1 (no epilogue), =2 (no contraction),
# 3 (no contraction, no am gathers), 4 (no contraction, no stores), 5 (no contraction, neither)
B=$PWD/tf-fast-rnnt_amd/csrc/_build
IFS=";" read -ra SHAPE_LIST <<< "${SHAPES:-32 2000 300 1024;32 1000 200 500}"
for shape in "${SHAPE_LIST[@]}"; do
  for ft in ${FTS:-64}; do
    for v in ${VARIANTS:-product fexp1 fexp2 fexp3 fexp4 fexp5}; do
      if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$B/libftr_$v.so; fi
      echo -n "frames=$ft $v: "; FTR_FUSED_FT=$ft python scripts/fused_bench.py $shape 2>&1 | grep -v amdgpu.ids | sed 's/.*: fused/fused/' | cut -c1-40
    done
  done
done
