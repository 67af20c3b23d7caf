#!/bin/bash
# round 4: LDS-staged stride-2 weight gradient -- tests, A/B timing, the training iteration
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_grad_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
SESSD_WGRAD_S2_LDS=0 timeout -k 5 300 python scripts/wgrad_s2_probe.py > $O/probe_old.json 2>$O/probe_old.err; echo "old rc $?"; cat $O/probe_old.json
timeout -k 5 300 python scripts/wgrad_s2_probe.py > $O/probe_new.json 2>$O/probe_new.err; echo "new rc $?"; cat $O/probe_new.json
timeout -k 5 600 python scripts/train_step_bench.py --real-loss --steps 20 > $O/train.json 2>$O/train.err; echo "train rc $?"; cut -c1-600 $O/train.json
