#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_seq
timeout -k 5 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_seq -o sq --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --cpu-frames 0 --no-roofline --streams 1 > $R/gpurun_out/prof_seq.log 2>&1
K=$(find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" | head -1)
python $R/scripts/frame_sequence.py $K | tee $R/gpurun_out/frame_sequence.txt
find $R/gpurun_out/prof_seq -name "*.csv" -size +1M -delete
