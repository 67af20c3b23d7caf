#!/bin/bash
# round 4: PMC passes of the sparse stage (batch 1 and the dense-scene batch), compact summary (scripts/pmc_compact.py)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4pmc
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
OUT=$R/gpurun_out/r4pmc/sparse_pmc_summary.txt
rm -f $OUT
for cfg in b1 stress; do
  flag=""; [ $cfg = stress ] && flag="--stress"
  files=""
  i=0
  for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D=$R/gpurun_out/r4pmc/${cfg}_pmc$i
    rm -rf $D
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py $flag --frames 3 > $R/gpurun_out/r4pmc/${cfg}_pmc$i.log 2>&1
    echo "$cfg pass $i rc $?"
    f=$(find $D -name "*counter_collection.csv" | head -1)
    files="$files $f"
    [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
  done
  T=420; [ $cfg = stress ] && T=420
  python $R/scripts/pmc_compact.py "SpMiddleFHD, $cfg" $files --trace $tr --tail $T --match sparse_conv --match chain_ --match vox_ >> $OUT
  grep "sites\|stages" $R/gpurun_out/r4pmc/${cfg}_pmc1.log | sed 's/^/# /' >> $OUT
  echo >> $OUT
  for i in 1 2 3; do rm -rf $R/gpurun_out/r4pmc/${cfg}_pmc$i; done
done
cat $OUT | cut -c1-170
