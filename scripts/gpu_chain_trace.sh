#!/bin/bash
# rocprofv3 kernel trace of the site-chain kernels alone: real frame / empty frame / real frame (5 runs each)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in b1 stress; do
  flag=""; [ $cfg = stress ] && flag="--stress"
  rm -rf $R/gpurun_out/chain_$cfg
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/chain_$cfg -o t -- python $R/scripts/chain_probe.py $flag > $R/gpurun_out/chain_$cfg.log 2>&1
  grep "^sites" $R/gpurun_out/chain_$cfg.log
  f=$(find $R/gpurun_out/chain_$cfg -name "*kernel_trace.csv" | head -1)
  python $R/scripts/chain_trace_summary.py $f
done
