#!/bin/bash
# round 6: engines per CU set re-measured with the round-6 list configuration (whole-unit shapes autotuned): 4 / 6 / 8 engines on halves, quarters
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h; mkdir -p $O $R/build
cd $R
W=build/r6_student.pt
[ -f $W ] || timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-host-io --no-sequential --no-roofline --cpu-frames 4 --save-weights $W > $O/train.json 2>$O/train.err
B="--weights $W --no-train-step --no-host-io --no-sequential --no-roofline --cpu-frames 8 --steps 600 --warmup 60"
run() { n=$1; shift; timeout -k 5 400 python bench.py $B "$@" > $O/ab_$n.json 2>$O/ab_$n.err; echo "$n rc $?"; }
run s4_p2_a --streams 4 --cu-parts 2
run s6_p2 --streams 6 --cu-parts 2
run s8_p2 --streams 8 --cu-parts 2
run s4_p4 --streams 4 --cu-parts 4
run s8_p4 --streams 8 --cu-parts 4
run s2_p2 --streams 2 --cu-parts 2
run s4_p2_b --streams 4 --cu-parts 2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6h/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f.split("/")[-1], round(d["value"], 1), c["frames_in_flight"], c["cu_sets"], c["cus_per_set"], round(c["ms_latency_per_frame_in_flight"], 3), c["parity_ok"], c["parity_matched"], c["parity_frames"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-300:])
PY
