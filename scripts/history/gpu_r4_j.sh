#!/bin/bash
# round 4: active-tile mode of the first SSFA layers -- tests + timing probe
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py tests/test_dense_conv_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log
timeout -k 5 600 python scripts/active_tiles_probe.py > $O/probe.json 2>$O/probe.err; echo "probe rc $?"; cat $O/probe.json; tail -3 $O/probe.err
