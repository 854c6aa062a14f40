# A/B of study builds of the recursion (csrc/mi_wave_bidir.hip) inside the bench step: product against _build/libftr_<name>.so
# for every name in VARIANTS (default: notouch = make -C tf-fast-rnnt_amd/csrc variant NAME=notouch SRC=mi_wave_bidir
# DEFS=-DFTR_EXP_NOTOUCH); interleaved, two rounds; columns: ms per step, roofline fraction of the pair, forward us, flow us
for round in 1 2; do
for v in ${VARIANTS:-notouch} product; do
  if [ $v = product ]; then unset FTR_LIB_PATH; else export FTR_LIB_PATH=$PWD/tf-fast-rnnt_amd/csrc/_build/libftr_$v.so; fi
  for cfg in ${CFGS:-c3 c4 c5 c2}; do
    python bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-dense --no-graph > gpurun_out/b12.json 2>gpurun_out/b12.err
    python -c "
import json; d=json.load(open('gpurun_out/b12.json')); e=d['roofline']['avg_launch_us_each']; print('$cfg $v', d['ms_per_step'], d['roofline']['frac'], round(e['ftr_mutual_information_fwd_ws_f32'],1), round(e['ftr_mutual_information_bwd_ws_f32'],1))"
  done
done; done
