#!/bin/bash
# round 4: active-tile mode in the engine -- tests, timing probe, bench A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_dense_active_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -12 $O/tests.log
timeout -k 5 600 python scripts/active_tiles_probe.py > $O/probe.json 2>$O/probe.err; echo "probe rc $?"; tail -2 $O/probe.err
for v in on off; do
  F=""; [ $v = off ] && F="--no-active-tiles"
  timeout -k 5 600 python bench.py --no-train-step $F > $O/bench_$v.json 2>$O/bench_$v.err; echo "bench $v rc $?"
  timeout -k 5 600 python bench.py --stress --no-train-step $F > $O/stress_$v.json 2>$O/stress_$v.err; echo "stress $v rc $?"
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4k/probe.json"))
for k,v in d.items():
    print(k, "fractions", [round(x,3) for x in v["active_tile_fraction"]], "activity %.1f us, fill %.1f us"%(v["activity_us"], v["fill_3_layers_us"]))
for n in ("bench_on","bench_off","stress_on","stress_off"):
    try:
        d=json.loads(open("gpurun_out/r4k/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"), round(d["roofline"]["frac"],3), d["roofline"].get("frac_full_map_launches"), d["stages_ms_eager"], d.get("sequential",{}).get("frames_per_s"), d["config"]["tuning"].get("active_tiles"), d["roofline"].get("active_tile_fraction"), d["roofline"]["dense_launch_ms"])
    except Exception as ex:
        print(n, "unreadable", ex)
PY
