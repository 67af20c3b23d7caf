#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_dense_conv_gpu.py -q --timeout 300 -x -k "stream_k" -s > gpurun_out/wino_sk_test.log 2>&1
grep -n "^E \|err \|passed\|failed" gpurun_out/wino_sk_test.log | cut -c1-200 | head -40
timeout -k 5 120 python scripts/wino_probe.py 21:0:200x176 22:0:200x176 23:0:200x176 22:256:200x176:2 22:256:200x176:10 22:248:200x176 22:240:200x176 2>&1 | tail -8
