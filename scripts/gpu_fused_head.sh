#!/bin/bash
# dense conv + pipeline tests, then the two batch-1 bench lines
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 900 python -m pytest tests/test_dense_conv_gpu.py tests/test_pipeline_gpu.py tests/test_forward_golden_gpu.py -m gpu -q -x --timeout 600 > gpurun_out/fh_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/fh_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
export SESSD_BENCH_VERBOSE=1
for s in 1 2; do
timeout -k 5 200 python -u bench.py --streams $s --cpu-frames 0 --no-host-io 2> gpurun_out/suite_bench_$s.err | tail -1 > gpurun_out/suite_bench_$s.json; python -c "
import json; d=json.load(open('gpurun_out/suite_bench_$s.json')); r=d['roofline']; print('streams $s', round(d['value'],1), round(d['ms_per_step'],4), d['stages_ms_eager']); print({k:(r['dense_tile_cfg'][k], round(v*1e3,1)) for k,v in r['dense_launch_ms'].items()})"
done
