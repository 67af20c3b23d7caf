#!/bin/bash
# round 4: counter passes of the dense stage with blocks 0 / 1 in active-tile mode (forced: see sparse_probe.py --force-active)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4dp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
files=""; i=0
for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$O/dense_pmc$i
  rm -rf $D
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --force-active > $O/dense_pmc$i.log 2>&1
  echo "dense pmc pass $i rc $?"
  f=$(find $D -name "*counter_collection.csv" | head -1)
  files="$files $f"
  [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
done
python $R/scripts/pmc_compact.py "SSFA neck + heads, batch 1, blocks 0 / 1 in active-tile mode" $files --trace $tr --tail 380 --match winograd --match conv2d_sk --match conv2d_mfma --match bev_tile --match fill_inactive --match ssfa_fuse > $O/dense_pmc_summary.txt
grep "active_tiles\|stages" $O/dense_pmc1.log | sed 's/^/# /' >> $O/dense_pmc_summary.txt
for i in 1 2 3; do rm -rf $O/dense_pmc$i; done
cut -c1-170 $O/dense_pmc_summary.txt | head -30
