#!/bin/bash
# PMC passes of one dense conv kernel alone: gpu_wino_pmc.sh "<probe args>" <tag> <kernel name substring>
# (PROBE=wino_probe.py by default; PASSES=4 by default, 2 = the SQ counter sets only)
set -u
R=$GRAFT_REPO_ROOT
ARGS=${1:-21}
TAG=${2:-wino}
KN=${3:-winograd}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
S1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU"
S2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
OUT=$R/gpurun_out/${TAG}_pmc_summary.txt
rm -f $OUT
i=0
for set in "$S1" "$S2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  [ $i -gt ${PASSES:-4} ] && break
  rm -rf $R/gpurun_out/${TAG}_pmc$i
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/${TAG}_pmc$i -o p --output-format csv -- python $R/scripts/${PROBE:-wino_probe.py} $ARGS > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
  echo "== pass $i rc=$? ($set)" >> $OUT
  f=$(find $R/gpurun_out/${TAG}_pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $KN >> $OUT
  k=$(find $R/gpurun_out/${TAG}_pmc$i -name "*kernel_trace.csv" | head -1)
  [ $i = 1 ] && [ -n "$k" ] && python - $k $KN >> $OUT <<'PY'
import csv, sys
v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
print("   duration_us(profiled) n=%d avg=%.1f min=%.1f" % (len(v), sum(v) / len(v), min(v)))
PY
  find $R/gpurun_out/${TAG}_pmc$i -name "*.csv" -size +2M -delete
done
cat $OUT
