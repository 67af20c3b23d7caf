#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 600 python -m pytest tests/test_pipeline_gpu.py -q --timeout 600 -s 2>&1 | tail -15 | cut -c1-300
B=8 python scripts/debug_stress_parity.py 2>&1 | grep -v "^   overlaps\|kept rows" | tail -12 | cut -c1-400
