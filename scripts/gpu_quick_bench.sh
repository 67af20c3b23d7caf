#!/bin/bash
# sparse / dense / pipeline tests, then the bench lines (1 stream, default 2 streams) with the tuning report
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 900 python -m pytest tests/test_datapath_gpu.py tests/test_di_nms_gpu.py tests/test_bn_train_gpu.py -q -x --timeout 600 > gpurun_out/quick_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/quick_tests.log | cut -c1-200; grep -n "^E " gpurun_out/quick_tests.log | head -10 | cut -c1-250
export SESSD_BENCH_VERBOSE=1
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2> gpurun_out/quick_1s.err | tail -1 > gpurun_out/quick_bench_1stream.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_1stream.json')); print('1stream', d['value'], d['ms_per_step'], d['stages_ms_eager']); r=d['roofline']; print('roofline', r['avg_launch_ms'], r['frac']); m=d['roofline_spmiddle']['mfma']; print('spmiddle mfma', m['conv_ms'], m['executed_tflops']); print([round(l['ms']*1e3,1) for l in m['layers']])"
grep -i "autotuned" gpurun_out/quick_1s.err | head -3 | cut -c1-900
timeout -k 5 300 python -u bench.py --cpu-frames 0 2> /dev/null | tail -1 > gpurun_out/quick_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_default.json')); print('default', d['value'], d['ms_per_step'])"
