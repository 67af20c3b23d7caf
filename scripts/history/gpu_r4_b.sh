#!/bin/bash
# round 4, second GPU call: the capacity-form head loss (tests + captured iteration + timing)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_head_loss_gpu.py tests/test_runner_gpu.py -x -q -m gpu > gpurun_out/r4b/tests_loss.log 2>&1
echo "loss tests rc $?"; tail -25 gpurun_out/r4b/tests_loss.log
timeout 1200 python -m pytest tests/test_train_gpu.py -q -m gpu -k "reference_loss or data_contract or captured" > gpurun_out/r4b/tests_train.log 2>&1
echo "train tests rc $?"; tail -25 gpurun_out/r4b/tests_train.log
timeout 600 python scripts/train_step_bench.py --real-loss --steps 20 > gpurun_out/r4b/train_step.json 2> gpurun_out/r4b/train_step.err
echo "train bench rc $?"; tail -c 600 gpurun_out/r4b/train_step.err; cat gpurun_out/r4b/train_step.json | cut -c1-1500
