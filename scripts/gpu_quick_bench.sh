#!/bin/bash
# dense + pipeline tests, then the bench lines (1 stream, default 2 streams) with the tuning report
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 900 python -m pytest tests/test_dense_conv_gpu.py tests/test_pipeline_gpu.py tests/test_forward_golden_gpu.py -q -x --timeout 600 > gpurun_out/quick_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/quick_tests.log | cut -c1-200
export SESSD_BENCH_VERBOSE=1
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2> gpurun_out/quick_1s.err | tail -1 > gpurun_out/quick_bench_1stream.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_1stream.json')); print('1stream', d['value'], d['ms_per_step'], d['stages_ms_eager']); r=d['roofline']; print('roofline', r['avg_launch_ms'], r['frac'], r['frac_algorithmic'], r.get('dense_launch_ms'))"
grep -i "tune\|tile_cfg" gpurun_out/quick_1s.err | head -5 | cut -c1-600
timeout -k 5 300 python -u bench.py --cpu-frames 0 2> /dev/null | tail -1 > gpurun_out/quick_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_default.json')); print('default', d['value'], d['ms_per_step'])"
