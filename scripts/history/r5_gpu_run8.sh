#!/bin/bash
# round 5 EXPERIMENT, fourth pass: four frames in flight on two CU halves -- persistent-launch size inside a half, mask layout, repeats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
run() { timeout -k 5 300 python bench.py --steps 800 --warmup 80 --cpu-frames 8 --no-host-io --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run s4_p2_a "--streams 4 --cu-split contiguous --cu-parts 2"
run s4_p2_b64 "--streams 4 --cu-split contiguous --cu-parts 2 --cu-budget 64"
run s4_p2_b96 "--streams 4 --cu-split contiguous --cu-parts 2 --cu-budget 96"
run s4_p2_inter "--streams 4 --cu-split interleaved --cu-parts 2"
run s4_p4_b "--streams 4 --cu-split contiguous --cu-parts 4"
run s4_p2_b "--streams 4 --cu-split contiguous --cu-parts 2"
run s3_p1_b88 "--streams 3 --cu-budget 88"
run s4_p1_mask "--streams 4 --cu-split contiguous --cu-parts 1 --cu-budget 64"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5k/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], d["config"].get("cus_per_frame_in_flight"), d["config"]["frames_in_flight"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-300:])
PY
