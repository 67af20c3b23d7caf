#!/bin/bash
# round 6: throughput-regime A/Bs of the Winograd list launches (shape / share rule chosen per launch in isolation vs forced), CU budgets,
# then the batch x CU-set sweep (review item 6)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d; mkdir -p $O $R/build
cd $R
W=build/r6_student.pt
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-host-io --no-sequential --no-roofline --cpu-frames 4 --save-weights $W > $O/train.json 2>$O/train.err; echo "train rc $?"
B="--weights $W --no-train-step --no-host-io --no-sequential --no-roofline --cpu-frames 8 --steps 600 --warmup 60"
run() { n=$1; shift; timeout -k 5 400 env "$@" python bench.py $B $EXTRA > $O/ab_$n.json 2>$O/ab_$n.err; echo "$n rc $?"; }
EXTRA="" run base_a X=1
EXTRA="" run shape0 SESSD_LIST_SHAPE=0
EXTRA="" run shape1 SESSD_LIST_SHAPE=1
EXTRA="" run shape0_b1 SESSD_LIST_SHAPE=0 SESSD_LIST_LAYERS=4,5
EXTRA="" run shape0_b0 SESSD_LIST_SHAPE=0 SESSD_LIST_LAYERS=0,1,2
EXTRA="" run two_units SESSD_LIST_MIN_ROUNDS=-2
EXTRA="" run shape0_two SESSD_LIST_SHAPE=0 SESSD_LIST_MIN_ROUNDS=-2
EXTRA="--cu-budget 112" run budget112 X=1
EXTRA="--cu-budget 120" run budget120 X=1
EXTRA="" run base_b X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6d/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]
        print(f.split("/")[-1], round(d["value"], 1), c.get("parity_ok"), c.get("parity_matched"), c.get("parity_frames"), c.get("parity_rule"), d["tuning"]["active_tiles"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-400:])
PY
WEIGHTS=$W bash scripts/r6_batch_cu_sweep.sh
