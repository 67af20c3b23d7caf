#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 500 python -m pytest tests/test_sparse_conv_gpu.py tests/test_sparse_grad_gpu.py tests/test_pipeline_gpu.py -q -x --timeout 300 2>&1 | tail -3
python scripts/sparse_layer_probe.py --layer 3 6 10 2>&1 | grep "^layer"
python scripts/sparse_layer_probe.py --stress --layer 3 6 10 --reps 10 2>&1 | grep "^layer"
cd /tmp && export TMPDIR=/tmp
for cfg in b1 stress; do
  flag=""; [ $cfg = stress ] && flag="--stress"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/lp_${cfg}_$C
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/lp_${cfg}_$C -o p --output-format csv -- python $R/scripts/sparse_layer_probe.py $flag --layer 6 --reps 5 > $R/gpurun_out/lp_${cfg}_$C.log 2>&1
    f=$(find $R/gpurun_out/lp_${cfg}_$C -name "*counter_collection.csv" | head -1)
    echo "== $cfg $C (layer 6: SubM 64->64)"; python - $f <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sparse_conv_kernel<64, 64" in r["Kernel_Name"]]
vals = [float(r["Counter_Value"]) for r in rows][-5:]
print("   last 5 launches:", vals)
PY
  done
  grep "^layer" $R/gpurun_out/lp_${cfg}_WRITE_SIZE.log
done
