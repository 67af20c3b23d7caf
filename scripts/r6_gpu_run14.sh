#!/bin/bash
# round 6 EXPERIMENT: one dense stage at a time per CU set (bench.py --dense-token): frames as two graphs, the back waits for the set's token
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O $R/build; cd $R
W=build/r6_student.pt
[ -f $W ] || timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-host-io --no-sequential --no-roofline --cpu-frames 4 --save-weights $W > $O/train.json 2>$O/train.err
B="--weights $W --no-train-step --no-host-io --no-sequential --no-roofline --cpu-frames 8 --steps 600 --warmup 60"
run() { n=$1; shift; timeout -k 5 400 python bench.py $B "$@" > $O/ab_$n.json 2>$O/ab_$n.err; echo "$n rc $?"; }
run base_a
run token_s4 --dense-token
run token_s6 --dense-token --streams 6
run token_s8 --dense-token --streams 8
run base_s6 --streams 6
run base_b
run token_s4_b --dense-token
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6m/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f.split("/")[-1], round(d["value"], 1), c["frames_in_flight"], c.get("dense_token"), round(c["ms_latency_per_frame_in_flight"], 3), c["parity_ok"], c["parity_matched"], c["parity_frames"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-500:])
PY
