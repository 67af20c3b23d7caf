#!/bin/bash
# two frames in flight: persistent workgroups of the stream-K launches (CUs left to the other stream's small kernels) vs frames/s
set -u
R=$GRAFT_REPO_ROOT
cd $R
for wg in ${WGS:-0 248 240 232 224}; do
  for st in ${STREAMS:-2}; do
    v=$(timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline --no-host-io --no-sequential --streams $st --sk-workgroups $wg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))")
    echo "streams $st sk_workgroups $wg -> $v"
  done
done
