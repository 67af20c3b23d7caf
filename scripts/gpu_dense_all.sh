#!/bin/bash
# all dense conv tests + the probe of the non-Winograd layers + both bench lines
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_conv_gpu.py tests/test_dense_grad_gpu.py -m gpu -q -x --timeout 300 > gpurun_out/dense_tests.log 2>&1
echo "tests exit $?"; tail -4 gpurun_out/dense_tests.log
timeout -k 5 200 python scripts/csk_probe.py 0 > gpurun_out/csk_probe.log 2>&1
echo "probe exit $?"; grep "cfg" gpurun_out/csk_probe.log | tail -40
export SESSD_BENCH_VERBOSE=1
for s in 1 2; do
timeout -k 5 200 python -u bench.py --streams $s --cpu-frames 0 --no-host-io 2> gpurun_out/csk_bench_$s.err | tail -1 > gpurun_out/csk_bench_$s.json; python -c "
import json; d=json.load(open('gpurun_out/csk_bench_$s.json')); r=d['roofline']; print('streams $s', round(d['value'],1), round(d['ms_per_step'],4), d['stages_ms_eager']); print({k:(r['dense_tile_cfg'][k], round(v*1e3,1)) for k,v in r['dense_launch_ms'].items()})"
done
