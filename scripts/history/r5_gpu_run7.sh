#!/bin/bash
# round 5 EXPERIMENT, third pass: several frames in flight PER CU set
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
run() { timeout -k 5 300 python bench.py --steps 800 --warmup 80 --cpu-frames 8 --no-host-io --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run s4_p4 "--streams 4 --cu-split contiguous --cu-parts 4"
run s8_p4 "--streams 8 --cu-split contiguous --cu-parts 4"
run s12_p4 "--streams 12 --cu-split contiguous --cu-parts 4"
run s4_p2 "--streams 4 --cu-split contiguous --cu-parts 2"
run s6_p2 "--streams 6 --cu-split contiguous --cu-parts 2"
run s8_p8 "--streams 8 --cu-split contiguous --cu-parts 8"
run s16_p8 "--streams 16 --cu-split contiguous --cu-parts 8"
run s8_p4_inter "--streams 8 --cu-split interleaved --cu-parts 4"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5j/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], d["config"].get("cus_per_frame_in_flight"), d["config"]["frames_in_flight"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-300:])
PY
