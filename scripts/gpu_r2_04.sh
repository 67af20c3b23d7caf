#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 500 python -m pytest tests/test_sparse_sites_gpu.py tests/test_pipeline_gpu.py -q -x --timeout 200 > gpurun_out/r2_04_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/r2_04_tests.log
bash scripts/gpu_r2_03.sh
