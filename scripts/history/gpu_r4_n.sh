#!/bin/bash
# round 4: frames in flight / stream-K workgroups with the active-tile dense stage (informational A/B on one box)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4n; mkdir -p $O
cd $R
for v in s2 s3 s2w224 s3w224 s2w192; do
  case $v in
    s2) F="--streams 2";; s3) F="--streams 3";; s2w224) F="--streams 2 --sk-workgroups 224";; s3w224) F="--streams 3 --sk-workgroups 224";; s2w192) F="--streams 2 --sk-workgroups 192";;
  esac
  timeout -k 5 300 python bench.py --no-train-step --no-host-io --no-sequential --no-roofline --cpu-frames 8 $F > $O/$v.json 2>$O/$v.err; echo "$v rc $?"
done
python - <<'PY'
import json
for n in ("s2","s3","s2w224","s3w224","s2w192"):
    try:
        d=json.loads(open("gpurun_out/r4n/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
