#!/bin/bash
# full GPU test suite + bench lines (default 2 streams, 1 stream, stress) + kernel trace of the 1-stream run
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2_07_tests.log 2>&1
echo "tests exit $?"; tail -4 gpurun_out/r2_07_tests.log
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/r2_07_bench.err | tail -1 > gpurun_out/r2_07_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r2_07_bench_default.json')); print('default', d['value'], d['ms_per_step'], d['stages_ms_eager'], d['roofline_spmiddle']['frac'], d.get('cpu_baseline',{}).get('value')); r=d['roofline']; print('roofline', r['avg_launch_ms'], r['frac'], r['frac_algorithmic']); m=d['roofline_spmiddle']['mfma']; print('spmiddle mfma', m['conv_ms'], m['executed_tflops'], m['executed_frac_of_f32_mfma_peak'], m['useful_row_fraction'])"
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/r2_07_bench_1stream.json; python -c "
import json; d=json.load(open('gpurun_out/r2_07_bench_1stream.json')); print('1stream', d['value'], d['ms_per_step'])"
timeout -k 5 400 python -u bench.py --stress --steps 30 --warmup 5 --cpu-frames 0 2> gpurun_out/r2_07_stress.err | tail -1 > gpurun_out/r2_07_bench_stress.json; python -c "
import json; d=json.load(open('gpurun_out/r2_07_bench_stress.json')); m=d['roofline_spmiddle'].pop('mfma'); print('stress', d['value'], d['ms_per_step'], d['stages_ms_eager'], d['roofline_spmiddle']); print('stress spmiddle mfma', m['conv_ms'], m['executed_tflops'], m['executed_frac_of_f32_mfma_peak'], m['useful_row_fraction'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r2c
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2c -o r2c -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 --no-roofline --no-host-io > $R/gpurun_out/prof_r2c.log 2>&1
DB=$(find $R/gpurun_out/prof_r2c -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 100 45 > $R/gpurun_out/prof_r2c_summary.txt; head -3 $R/gpurun_out/prof_r2c_summary.txt | cut -c1-150
find $R/gpurun_out/prof_r2c -name "*.db" -delete
rm -rf $R/gpurun_out/prof_r2d
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2d -o r2d -- python $R/bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-roofline --no-host-io > $R/gpurun_out/prof_r2d.log 2>&1
DB=$(find $R/gpurun_out/prof_r2d -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 200 45 > $R/gpurun_out/prof_r2d_summary.txt; head -3 $R/gpurun_out/prof_r2d_summary.txt | cut -c1-150
find $R/gpurun_out/prof_r2d -name "*.db" -delete
cd $R
