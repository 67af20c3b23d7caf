#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
timeout -k 5 600 python scripts/active_tiles_probe.py > $O/probe.json 2>$O/probe.err; echo "probe rc $?"; cat $O/probe.json; tail -3 $O/probe.err
