# fused forward: the product library against a build of another revision of simple_fused.hip (_build/libftr_fhead.so)
B=$PWD/tf-fast-rnnt_amd/csrc/_build
IFS=";" read -ra SHAPE_LIST <<< "${SHAPES:-32 1000 200 500;32 512 100 500;8 8000 1000 512;32 2000 300 1024}"
for shape in "${SHAPE_LIST[@]}"; do
  for v in product fhead product fhead; do
    if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$B/libftr_$v.so; fi
    echo -n "$shape $v: "; python scripts/fused_bench.py $shape 2>&1 | grep -v amdgpu.ids | sed 's/.*: fused/fused/' | cut -c1-40
  done
done
