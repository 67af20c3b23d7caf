#!/bin/bash
# round 5 EXPERIMENT, second pass: how many frames in flight, masked CU sets against persistent launches merely SIZED for a share
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
run() { timeout -k 5 300 python bench.py --steps 600 --warmup 60 --cpu-frames 8 --no-host-io --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run s3_mask "--streams 3 --cu-split contiguous"
run s4_mask "--streams 4 --cu-split contiguous"
run s5_mask "--streams 5 --cu-split contiguous"
run s6_mask "--streams 6 --cu-split contiguous"
run s3_budget "--streams 3 --cu-budget 80"
run s4_budget "--streams 4 --cu-budget 64"
run s4_budget96 "--streams 4 --cu-budget 96"
run s4_budget128 "--streams 4 --cu-budget 128"
run s6_budget "--streams 6 --cu-budget 40"
run s8_budget64 "--streams 8 --cu-budget 64"
run s4_mask_b "--streams 4 --cu-split contiguous"
run s2_base "--streams 2"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5i/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], d["config"].get("cus_per_frame_in_flight"), d["config"]["frames_in_flight"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-300:])
PY
