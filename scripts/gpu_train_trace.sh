#!/bin/bash
# training iteration: time + kernel trace
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 300 python scripts/train_step_bench.py --steps 10 2> gpurun_out/train_step.err | tail -1 > gpurun_out/train_step.json; cut -c1-400 gpurun_out/train_step.json; tail -3 gpurun_out/train_step.err | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o tr -- python $R/scripts/train_step_bench.py --steps 10 > $R/gpurun_out/prof_train.log 2>&1
DB=$(find $R/gpurun_out/prof_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 0 60 > $R/gpurun_out/train_trace_summary.txt; head -45 $R/gpurun_out/train_trace_summary.txt | cut -c1-175
find $R/gpurun_out/prof_train -name "*.db" -delete
