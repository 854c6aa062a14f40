# simple_bwd_am_kernel: product against study builds (prefetched frames / workgroups per CU) and the previous revision
B=$PWD/tf-fast-rnnt_amd/csrc/_build
IFS=";" read -ra SHAPE_LIST <<< "${SHAPES:-32 1000 200 500;32 512 100 500;8 8000 1000 512;32 2000 300 1024}"
for shape in "${SHAPE_LIST[@]}"; do
  for v in ${VARIANTS:-product pf6 pf8w3 pf8w2 fhead product fhead}; do
    if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$B/libftr_$v.so; fi
    echo -n "$v: "; python scripts/bwd_am_bench.py $shape 2>&1 | grep -v amdgpu.ids
  done
done
