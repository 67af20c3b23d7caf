#!/bin/bash
# A/B of autotune candidate sets inside ONE box: frames/s with one and two frames in flight
set -u
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for flags in "" "--no-offset-split" "--no-streamk" ; do
  for st in 1 2; do
    timeout -k 5 200 python -u bench.py --cpu-frames 0 --no-roofline --streams $st $flags 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rep $rep streams $st flags [$flags]:', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
done
