#!/bin/bash
# round 6: conv_0 / conv_1 over their tile list (step program {.., 3, 4, 0}): kernel tests, pipeline tests, smoke, A/B in the timed regime
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i; mkdir -p $O $R/build
cd $R
timeout -k 5 1800 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -12 $O/tests.log | cut -c1-300
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-400
W=build/r6_student.pt
[ -f $W ] || timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-host-io --no-sequential --no-roofline --cpu-frames 4 --save-weights $W > $O/train.json 2>$O/train.err
B="--weights $W --no-train-step --no-host-io --no-sequential --cpu-frames 16 --steps 600 --warmup 60"
run() { n=$1; shift; timeout -k 5 400 python bench.py $B "$@" > $O/ab_$n.json 2>$O/ab_$n.err; echo "$n rc $?"; }
run conv_list_a
run conv_full_a --no-active-conv
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6i/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
        print(f.split("/")[-1], round(d["value"], 1), c["parity_ok"], c["parity_matched"], c["parity_frames"], c["parity_rule"], round(r["frac"], 4), round(r["frac_chip_timed_region"], 4),
              {k: round(v * 1e3, 1) for k, v in r["dense_launch_ms"].items()}, {k: round(v, 3) for k, v in r["active_tile_fraction"].items()}, d["tuning"]["active_tiles"].get("conv_0+conv_1"))
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-600:])
PY
