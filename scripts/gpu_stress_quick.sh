#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 400 python -u bench.py --stress --steps 20 --warmup 5 --cpu-frames 0 2> gpurun_out/stress_quick.err | tail -1 > gpurun_out/stress_quick.json; python -c "
import json; d=json.load(open('gpurun_out/stress_quick.json')); m=d['roofline_spmiddle'].pop('mfma'); print('stress', d['value'], d['ms_per_step'], d['stages_ms_eager']); print('stress spmiddle mfma', m['conv_ms'], m['executed_tflops'], m['executed_frac_of_f32_mfma_peak'], m['useful_row_fraction']); print([round(l['ms']*1e3,1) for l in m['layers']])"
grep -i "autotuned" gpurun_out/stress_quick.err | head -3 | cut -c1-700
