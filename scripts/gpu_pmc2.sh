R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]+|TA_[A-Za-z_0-9]+|TCP_[A-Za-z_0-9]+|TD_[A-Za-z_0-9]+)" | sort -u > $R/gpurun_out/counters.txt
wc -l $R/gpurun_out/counters.txt
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAVES -d $R/gpurun_out/pmc3 -o p3 --output-format csv -- python $R/scripts/conv_occupancy_probe2.py > $R/gpurun_out/pmc3.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $R/gpurun_out/pmc4 -o p4 --output-format csv -- python $R/scripts/conv_occupancy_probe2.py > $R/gpurun_out/pmc4.log 2>&1
tail -2 $R/gpurun_out/pmc4.log
