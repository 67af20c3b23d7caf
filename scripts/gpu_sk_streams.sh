#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "2 224" "2 128" "2 120" "2 104" "3 80" "2 160"; do
  set -- $cfg
  timeout -k 5 200 python -u bench.py --cpu-frames 0 --no-roofline --streams $1 --sk-workgroups $2 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 streams, sk workgroups $2:', round(d['value'],1), round(d['ms_per_step'],4))"
done
