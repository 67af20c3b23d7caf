#!/bin/bash
# frames/s of the multi-stream bench against the stream-K shape / number of persistent workgroups: "streams cfg workgroups" triples
set -u
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "2 22 128" "2 22 120" "2 22 136" "2 22 144" "2 22 112" "3 22 80" "3 22 88" "2 22 160"; do
  set -- $cfg
  timeout -k 5 200 python -u bench.py --cpu-frames 0 --no-roofline --streams $1 --wino-cfg $2 --sk-workgroups $3 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 streams, wino cfg $2, sk workgroups $3:', round(d['value'],1), round(d['ms_per_step'],4))"
done
