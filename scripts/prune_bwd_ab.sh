# A/B of do_pruning_bwd: VARIANTS = default | segN (FTR_PRUNE_SEG=N frames per segment) | NAME of a study build
# (make -C tf-fast-rnnt_amd/csrc variant NAME=... SRC=prune DEFS=...); per-kernel times by rocprofv3
R=${GRAFT_REPO_ROOT:-$PWD}
B=$R/tf-fast-rnnt_amd/csrc/_build
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-c3}; do
  for v in ${VARIANTS:-default}; do
    unset FTR_LIB_PATH FTR_PRUNE_SEG
    case $v in default) ;; seg*) export FTR_PRUNE_SEG=${v#seg} ;; *) export FTR_LIB_PATH=$B/libftr_$v.so ;; esac
    rm -rf $R/gpurun_out/prof_prune_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prune_$v -- python3 $R/scripts/prune_bwd_study.py $cfg > $R/gpurun_out/prune_prof_$v.log 2>&1
    echo "== $cfg $v"; grep "do_pruning_bwd\|err" $R/gpurun_out/prune_prof_$v.log
    f=$(find $R/gpurun_out/prof_prune_$v -name "*kernel_stats.csv" | head -1)
    grep "do_pruning_bwd" $f | awk -F'",' '{split($2,a,","); n=split($1,nm,"::"); print "   ", substr(nm[n],1,40), a[3]/1000 " us avg"}'
  done
done
