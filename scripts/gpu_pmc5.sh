R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc6 -o p6 --output-format csv -- python $R/scripts/conv_occupancy_probe2.py > $R/gpurun_out/pmc6.log 2>&1
tail -1 $R/gpurun_out/pmc6.log | cut -c1-200
