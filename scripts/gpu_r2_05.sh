#!/bin/bash
# PMC passes of the chain kernels (chain_probe.py, batch 1)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
i=0
for set in "$SQ1" "$SQ2"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/chain_pmc$i
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/chain_pmc$i -o p --output-format csv -- python $R/scripts/chain_probe.py > $R/gpurun_out/chain_pmc$i.log 2>&1
  f=$(find $R/gpurun_out/chain_pmc$i -name "*counter_collection.csv" | head -1)
  python - $f <<'PY'
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(chain_\w+)", row["Kernel_Name"])
    if m:
        acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        # launches come as 5 real, 5 empty, 5 real (gather: x3 levels): print the first five (real frame)
        print("   %-24s" % c, " ".join("%9.3g" % x for x in v[:6]))
PY
done
