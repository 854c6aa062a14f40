mkdir -p gpurun_out
timeout -k 10 300 python scripts/mi_bench.py > gpurun_out/mi_bench.log 2>&1; echo "exit $?" >> gpurun_out/mi_bench.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/scripts/mi_bench.py 32 200 1000 > $R/gpurun_out/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/scripts/mi_bench.py 32 200 1000 > $R/gpurun_out/pmc2.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/scripts/mi_bench.py 32 200 1000 > $R/gpurun_out/pmc3.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/scripts/mi_bench.py 32 200 1000 > $R/gpurun_out/pmc4.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc5 -- python3 $R/scripts/mi_bench.py 32 200 1000 > $R/gpurun_out/pmc5.log 2>&1
cd $R; cat gpurun_out/mi_bench.log; ls gpurun_out/pmc*/ 2>/dev/null | head; tail -3 gpurun_out/pmc3.log
