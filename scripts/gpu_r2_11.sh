#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 500 python -m pytest tests/test_sparse_conv_gpu.py tests/test_pipeline_gpu.py -q -x --timeout 300 2>&1 | tail -2
python scripts/sparse_layer_probe.py --layer 3 6 10 2>&1 | grep "^layer" | cut -c1-120
python scripts/sparse_layer_probe.py --stress --layer 1 3 6 10 --reps 10 2>&1 | grep "^layer" | cut -c1-120
