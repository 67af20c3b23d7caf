#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 5 900 python -m pytest tests/test_kitti_eval_gpu.py tests/test_datapath_gpu.py -q --timeout 600 -x > gpurun_out/new_tests.log 2>&1
grep -n "^E " gpurun_out/new_tests.log | cut -c1-300 | head -30; tail -5 gpurun_out/new_tests.log | cut -c1-300
