#!/bin/bash
# PMC passes of the sparse stage with the round-2 kernels (batch 1 and dense-scene batch 8)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD"
rm -f $R/gpurun_out/sp3_pmc_summary.txt
for cfg in b1 stress; do
  flag=""; [ $cfg = stress ] && flag="--stress"
  i=0
  for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/sp3_${cfg}_pmc$i
    timeout -k 5 240 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/sp3_${cfg}_pmc$i -o p --output-format csv -- python $R/scripts/sparse_probe.py $flag --frames 3 > $R/gpurun_out/sp3_${cfg}_pmc$i.log 2>&1
    f=$(find $R/gpurun_out/sp3_${cfg}_pmc$i -name "*counter_collection.csv" | head -1)
    echo "== $cfg pass $i" >> $R/gpurun_out/sp3_pmc_summary.txt
    python $R/scripts/pmc_summary.py $f sparse_conv_kernel >> $R/gpurun_out/sp3_pmc_summary.txt
    python $R/scripts/pmc_summary.py $f chain_ >> $R/gpurun_out/sp3_pmc_summary.txt
    # per-launch durations of the same pass, for executed-FLOP rates
    k=$(find $R/gpurun_out/sp3_${cfg}_pmc$i -name "*kernel_trace.csv" | head -1)
    [ $i = 1 ] && python - $k >> $R/gpurun_out/sp3_pmc_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "sparse_conv_kernel" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:86]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print("   duration_us(profiled) %-86s n=%3d avg=%.1f" % (k, len(v), sum(v) / len(v)))
PY
  done
  grep "sites\|tuning" $R/gpurun_out/sp3_${cfg}_pmc1.log
done
wc -l $R/gpurun_out/sp3_pmc_summary.txt
