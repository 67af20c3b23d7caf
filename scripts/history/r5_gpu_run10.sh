#!/bin/bash
# round 5: the host-fed pipeline with four engines on CU sets -- why 1237 against 1405 with two plain streams?
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
run() { SESSD_HOSTIO=$3 timeout -k 5 300 python bench.py --steps 100 --warmup 20 --cpu-frames 4 --no-sequential --no-train-step --no-roofline $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run d_16_4 "" 16,4
run d_32_4 "" 32,4
run d_16_8 "" 16,8
run d_8_4 "" 8,4
run d_64_8 "" 64,8
run s2split_16_4 "--streams 2" 16,4
run s4none_16_4 "--cu-split none" 16,4
run s2none_16_4 "--streams 2 --cu-split none" 16,4
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5m/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d.get("host_io") or {}
        print(f.split("/")[-1], round(d["value"], 1), round(h.get("frames_per_s", 0), 1), round(h.get("latency_mode_frames_per_s", 0), 1), h.get("last_frame_equals_latency_mode"))
    except Exception as ex:
        print(f, "unreadable", ex)
PY
