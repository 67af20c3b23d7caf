#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 300 python -m pytest tests/test_dense_grad_gpu.py -q -x --timeout 300 2>&1 | tail -2
timeout -k 5 120 python scripts/wgrad_probe.py 2>&1 | tail -5
