timeout 120 python scripts/conv_microbench.py 20 2>&1 | tail -24
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc1 -o p1 --output-format csv -- python $R/scripts/conv_microbench.py 3 1 > $R/gpurun_out/pmc1.log 2>&1
tail -3 $R/gpurun_out/pmc1.log; ls $R/gpurun_out/pmc1
timeout 100 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc2 -o p2 --output-format csv -- python $R/scripts/conv_microbench.py 3 1 > $R/gpurun_out/pmc2.log 2>&1
tail -3 $R/gpurun_out/pmc2.log
