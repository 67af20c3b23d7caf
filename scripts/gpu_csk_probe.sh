#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 200 python scripts/csk_probe.py ${CSK_WGS:-0} > gpurun_out/csk_probe.log 2>&1
echo "probe exit $?"; grep "cfg 30" gpurun_out/csk_probe.log | tail -40
