#!/bin/bash
# round 4, third GPU call: SyncBN + two-rank tests, reproducibility probe, full train test file
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4c
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_bn_train_gpu.py -q -m gpu -x > gpurun_out/r4c/tests_train.log 2>&1
echo "train tests rc $?"; tail -30 gpurun_out/r4c/tests_train.log
timeout 600 python scripts/repro_probe.py > gpurun_out/r4c/repro_real.json 2> gpurun_out/r4c/repro_real.err
echo "repro real rc $?"; tail -c 400 gpurun_out/r4c/repro_real.err; cut -c1-2500 gpurun_out/r4c/repro_real.json
timeout 600 python scripts/repro_probe.py --standin > gpurun_out/r4c/repro_standin.json 2> gpurun_out/r4c/repro_standin.err
echo "repro standin rc $?"; cut -c1-2500 gpurun_out/r4c/repro_standin.json
