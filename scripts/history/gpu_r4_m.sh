#!/bin/bash
# round 4: active tiles on the LDS-tiled stream-K kernel (b1.0, trans_0, trans_1) -- tests, bench
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py -x -q -m gpu > $O/tests_active.log 2>&1; echo "active tests rc $?"; tail -15 $O/tests_active.log
timeout -k 5 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "active_tiles or whatever_autotune or stress_autotuned" > $O/tests_pipe.log 2>&1; echo "pipeline tests rc $?"; tail -12 $O/tests_pipe.log
timeout -k 5 600 python bench.py --no-train-step > $O/bench_on.json 2>$O/bench_on.err; echo "bench rc $?"; tail -3 $O/bench_on.err
timeout -k 5 600 python bench.py --stress --no-train-step --no-host-io > $O/stress_on.json 2>$O/stress_on.err; echo "stress rc $?"
python - <<'PY'
import json
for n in ("bench_on","stress_on"):
    try:
        d=json.loads(open("gpurun_out/r4m/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"), round(d["roofline"]["frac"],3), d["roofline"].get("frac_full_map_launches"), d["stages_ms_eager"], d.get("value_sequential",{}).get("frames_per_s"), d["config"]["tuning"].get("active_tiles"), d["roofline"].get("active_tile_fraction"), d["roofline"]["dense_launch_ms"], d["config"]["tuning"]["dense_tile_cfg"])
    except Exception as ex:
        print(n, "unreadable", ex)
PY
