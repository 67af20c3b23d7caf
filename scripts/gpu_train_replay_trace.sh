#!/bin/bash
# kernel table of the CAPTURED training iteration alone: one eager warm-up iteration, the capture pass (not executed) and 40 replays
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_replay
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_replay -o tr -- python $R/scripts/train_step_bench.py --replays-only 40 > $R/gpurun_out/prof_replay.log 2>&1
tail -1 $R/gpurun_out/prof_replay.log | cut -c1-300
DB=$(find $R/gpurun_out/prof_replay -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 40 140 sparse_pack_batch_kernel > $R/gpurun_out/train_replay_trace.txt; head -3 $R/gpurun_out/train_replay_trace.txt | cut -c1-150; tail -1 $R/gpurun_out/train_replay_trace.txt
rm -rf $R/gpurun_out/prof_replay
