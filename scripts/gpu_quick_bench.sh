#!/bin/bash
# the bench lines (1 stream, default 2 streams, optionally --stress with "stress" as $1) with the autotune report; no tests
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2> gpurun_out/quick_1s.err | tail -1 > gpurun_out/quick_bench_1stream.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_1stream.json')); print('1stream', d['value'], d['ms_per_step'], d['stages_ms_eager']); r=d['roofline']; print('roofline', r['avg_launch_ms'], r['frac']); m=d['roofline_spmiddle']['mfma']; print('spmiddle mfma', m['conv_ms'], m['executed_tflops']); print([round(l['ms']*1e3,1) for l in m['layers']])"
grep -i "autotuned" gpurun_out/quick_1s.err | head -3 | cut -c1-900
timeout -k 5 300 python -u bench.py --cpu-frames 0 2> /dev/null | tail -1 > gpurun_out/quick_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_default.json')); print('default', d['value'], d['ms_per_step'])"
if [ "${1:-}" = stress ]; then
timeout -k 5 400 python -u bench.py --stress --steps 20 --warmup 5 --cpu-frames 0 2> gpurun_out/quick_stress.err | tail -1 > gpurun_out/quick_bench_stress.json; python -c "
import json; d=json.load(open('gpurun_out/quick_bench_stress.json')); m=d['roofline_spmiddle'].pop('mfma'); print('stress', d['value'], d['ms_per_step'], d['stages_ms_eager']); print('stress spmiddle mfma', m['conv_ms'], m['executed_tflops'], m['executed_frac_of_f32_mfma_peak'], m['useful_row_fraction'])"
fi
