#!/bin/bash
# Round 2, GPU call 1: (a) validate the kernels written without GPU access in round 1, (b) PMC passes of the sparse stage.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
export SESSD_EXPERIMENTAL=1
timeout -k 5 400 python -m pytest tests/test_site_renumber_gpu.py tests/test_datapath_gpu.py tests/test_sparse_conv_deep_gpu.py tests/test_bn_train_gpu.py -q -x --timeout 120 > gpurun_out/experimental_tests.log 2>&1
echo "experimental tests exit $?" | tee -a gpurun_out/experimental_tests.log
tail -25 gpurun_out/experimental_tests.log
unset SESSD_EXPERIMENTAL
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"
for cfg in b1 stress; do
  flag=""; [ $cfg = stress ] && flag="--stress"
  timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sp_${cfg}_trace -o t -- python $R/scripts/sparse_probe.py $flag --frames 6 > $R/gpurun_out/sp_${cfg}_trace.log 2>&1
  tail -3 $R/gpurun_out/sp_${cfg}_trace.log
  i=0
  for set in "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 5 240 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/sp_${cfg}_pmc$i -o p --output-format csv -- python $R/scripts/sparse_probe.py $flag --frames 3 > $R/gpurun_out/sp_${cfg}_pmc$i.log 2>&1
    f=$(find $R/gpurun_out/sp_${cfg}_pmc$i -name "*counter_collection.csv" | head -1)
    echo "== $cfg pass $i: $f" >> $R/gpurun_out/sp_pmc_summary.txt
    python $R/scripts/pmc_summary.py $f sparse_conv_kernel >> $R/gpurun_out/sp_pmc_summary.txt
    python $R/scripts/pmc_summary.py $f rulebook_kernel >> $R/gpurun_out/sp_pmc_summary.txt
    python $R/scripts/pmc_summary.py $f down_insert >> $R/gpurun_out/sp_pmc_summary.txt
  done
  DB=$(find $R/gpurun_out/sp_${cfg}_trace -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB 4 40 > $R/gpurun_out/sp_${cfg}_trace_summary.txt
  # keep the merged-back payload small
  find $R/gpurun_out/sp_${cfg}_trace $R/gpurun_out/sp_${cfg}_pmc* -name "*.db" -delete
done
cd $R
timeout -k 5 300 python scripts/renumber_bench.py > gpurun_out/renumber_bench.log 2>&1
tail -6 gpurun_out/renumber_bench.log
wc -c gpurun_out/sp_pmc_summary.txt
