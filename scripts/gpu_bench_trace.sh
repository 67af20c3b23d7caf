#!/bin/bash
# the default bench line (2 frames in flight) and the rocprofv3 kernel trace of the same command (no roofline / cpu legs in the trace)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/r2_08_bench.err | tail -1 > gpurun_out/r2_08_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r2_08_bench_default.json')); print('default', d['value'], d['ms_per_step'], d['stages_ms_eager'], d['config']); r=d['roofline']; print('roofline', r['avg_launch_ms'], r['frac'], r['frac_algorithmic'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r2e
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2e -o r2e -- python $R/bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-roofline --no-host-io > $R/gpurun_out/prof_r2e.log 2>&1
DB=$(find $R/gpurun_out/prof_r2e -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 200 45 > $R/gpurun_out/prof_r2e_summary.txt; head -8 $R/gpurun_out/prof_r2e_summary.txt | cut -c1-150
find $R/gpurun_out/prof_r2e -name "*.db" -delete
cd $R
