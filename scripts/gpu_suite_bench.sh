#!/bin/bash
# full GPU test suite + the two batch-1 bench lines (no traces)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/suite_tests.log 2>&1
echo "tests exit $?"; tail -4 gpurun_out/suite_tests.log
export SESSD_BENCH_VERBOSE=1
for s in 1 2; do
timeout -k 5 200 python -u bench.py --streams $s --cpu-frames 0 --no-host-io 2> gpurun_out/suite_bench_$s.err | tail -1 > gpurun_out/suite_bench_$s.json; python -c "
import json; d=json.load(open('gpurun_out/suite_bench_$s.json')); r=d['roofline']; print('streams $s', round(d['value'],1), round(d['ms_per_step'],4), d['stages_ms_eager'], 'roofline', round(r['avg_launch_ms']*1e3,1), round(r['frac'],3)); print({k:(r['dense_tile_cfg'][k], round(v*1e3,1)) for k,v in r['dense_launch_ms'].items()}); m=d['roofline_spmiddle']['mfma']; print('spmiddle convs', m['conv_ms'], m['executed_tflops'])"
done
