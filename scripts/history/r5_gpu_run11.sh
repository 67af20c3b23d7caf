#!/bin/bash
# round 5: knobs under the CU-set default (list-share modes, active tiles off, repeats)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5p; mkdir -p $O
cd $R
run() { timeout -k 5 300 python bench.py --steps 800 --warmup 80 --cpu-frames 8 --no-host-io --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run def_a ""
run cut "--list-shares cut"
run whole "--list-shares whole"
run noactive "--no-active-tiles"
run def_b ""
run nostreamk "--no-streamk"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5p/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], d["config"]["tuning"]["active_tiles"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-300:])
PY
