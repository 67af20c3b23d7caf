#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
for wg in 0 248 240 224 192; do
  timeout -k 5 200 python -u bench.py --cpu-frames 0 --no-roofline --sk-workgroups $wg 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 streams, sk workgroups $wg:', round(d['value'],1), round(d['ms_per_step'],4))"
done
timeout -k 5 200 python -u bench.py --cpu-frames 0 --no-roofline --streams 3 --sk-workgroups 224 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('3 streams, sk workgroups 224:', round(d['value'],1), round(d['ms_per_step'],4))"
