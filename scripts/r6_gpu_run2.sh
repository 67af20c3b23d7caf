#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b; mkdir -p $O $R/build
cd $R
timeout -k 5 600 python scripts/r6_sparse_overflow_scan.py 64 300 > $O/overflow_scan.log 2>&1; echo "scan rc $?"; grep -v Warning $O/overflow_scan.log | tail -6 | cut -c1-900
timeout -k 5 900 python -m pytest tests/test_rccl_gpu.py -q -m gpu -x > $O/rccl.log 2>&1; echo "rccl rc $?"; tail -5 $O/rccl.log | cut -c1-1500
