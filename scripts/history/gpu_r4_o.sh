#!/bin/bash
# round 4: kernel trace of the sequential frame with the active-tile dense stage
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--steps 100 --warmup 10 --streams 1"; F=100
rm -rf $O/p
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p -o t -- python $R/bench.py $A --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p.log 2>&1
echo "rc $?"
DB=$(find $O/p -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $F 60 > $O/trace_1stream.txt; head -50 $O/trace_1stream.txt | cut -c1-175
rm -rf $O/p
