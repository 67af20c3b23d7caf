#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5q; mkdir -p $O
cd $R
run() { timeout -k 5 300 python bench.py --steps 800 --warmup 80 --cpu-frames 8 --no-host-io --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run whole_a "--list-shares whole"
run auto_a ""
run whole_b "--list-shares whole"
run auto_b ""
run whole_c "--list-shares whole"
run whole_p4 "--list-shares whole --cu-parts 4"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5q/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"])
    except Exception as ex:
        print(f, "unreadable", ex)
PY
