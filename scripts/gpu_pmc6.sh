R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc7a -o p --output-format csv -- python $R/scripts/wino_probe.py 20 21 > $R/gpurun_out/pmc7a.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmc7b -o p --output-format csv -- python $R/scripts/wino_probe.py 20 > $R/gpurun_out/pmc7b.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc7c -o p --output-format csv -- python $R/scripts/wino_probe.py 20 > $R/gpurun_out/pmc7c.log 2>&1
for d in pmc7a pmc7b pmc7c; do f=$(find $R/gpurun_out/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; python $R/scripts/pmc_summary.py $f winograd; done
