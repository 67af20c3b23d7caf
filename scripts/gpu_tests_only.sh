#!/bin/bash
# the driver's GPU tier: every GPU-marked test
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_tests.log 2>&1
echo "tests exit $?"; tail -8 gpurun_out/gpu_tests.log | cut -c1-300
