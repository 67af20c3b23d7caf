#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
timeout -k 5 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-roofline --no-host-io > $R/gpurun_out/prof_tl.log 2>&1
K=$(find $R/gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
head -1 $K | cut -c1-400
python $R/scripts/timeline_overlap.py $K 150 | tee $R/gpurun_out/timeline_2streams.txt
find $R/gpurun_out/prof_tl -name "*.csv" -size +1M -delete
