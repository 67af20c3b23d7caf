#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5r; mkdir -p $O
cd $R
run() { timeout -k 5 300 python bench.py --steps 800 --warmup 80 --cpu-frames 8 --no-host-io --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run def_a ""
run s6_p2 "--streams 6"
run s3_p2 "--streams 3"
run no_offset_split "--no-offset-split"
run budget120 "--cu-budget 120"
run def_b ""
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5r/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], d["config"]["tuning"]["sparse"])
    except Exception as ex:
        print(f, "unreadable", ex)
PY
