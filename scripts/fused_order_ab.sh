# tile order of the fused forward with several symbol tiles per utterance: product against the FTR_EXP_FUSED_ORDER_OLD build
B=$PWD/tf-fast-rnnt_amd/csrc/_build
IFS=";" read -ra SHAPE_LIST <<< "${SHAPES:-32 2000 300 1024;8 8000 1000 512;32 1500 400 500;16 3000 600 768}"
for shape in "${SHAPE_LIST[@]}"; do
  for v in product forderold product forderold; do
    if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$B/libftr_$v.so; fi
    echo -n "$shape $v: "; python scripts/fused_bench.py $shape 2>&1 | grep -v amdgpu.ids | sed 's/.*: fused/fused/' | cut -c1-150
  done
done
