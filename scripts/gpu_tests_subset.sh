#!/bin/bash
# sparse / dense / pipeline tests, then the bench lines (1 stream, default 2 streams) with the tuning report
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 900 python -m pytest tests/test_dense_conv_gpu.py tests/test_bn_train_gpu.py tests/test_train_gpu.py tests/test_dense_grad_gpu.py tests/test_pipeline_gpu.py tests/test_forward_golden_gpu.py -q -x --timeout 600 > gpurun_out/quick_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/quick_tests.log | cut -c1-200; grep -n "^E " gpurun_out/quick_tests.log | head -10 | cut -c1-250
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python scripts/train_step_bench.py --steps 10 2> gpurun_out/train_step.err | tail -1 > gpurun_out/train_step.json; cut -c1-300 gpurun_out/train_step.json; tail -3 gpurun_out/train_step.err | cut -c1-300
