#!/bin/bash
# sparse / dense / pipeline tests, then the bench lines (1 stream, default 2 streams) with the tuning report
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 900 python -m pytest tests/test_datapath_gpu.py tests/test_di_nms_gpu.py tests/test_bn_train_gpu.py -q -x --timeout 600 > gpurun_out/quick_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/quick_tests.log | cut -c1-200; grep -n "^E " gpurun_out/quick_tests.log | head -10 | cut -c1-250
export SESSD_BENCH_VERBOSE=1
