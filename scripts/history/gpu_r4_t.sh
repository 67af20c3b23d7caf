#!/bin/bash
# round 4: fill launch with one mask test per 16 channels -- list tests, the engine's active-tile test, smoke, bench with its gate
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4t; mkdir -p $O
cd $R
timeout -k 5 100 python -m pytest tests/test_dense_active_gpu.py tests/test_site_renumber_gpu.py -x -q -m gpu > $O/tests_active.log 2>&1; echo "active tests rc $?"; tail -3 $O/tests_active.log
timeout -k 5 100 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "active_tiles or stress_autotuned" > $O/tests_pipe.log 2>&1; echo "pipeline tests rc $?"; tail -3 $O/tests_pipe.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout -k 5 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-train-step --no-host-io > $O/bench_driver.json 2>$O/bench_driver.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4t/bench_driver.json").read().strip().splitlines()[-1])
    print(round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"), d["parity"].get("frames"), d["stages_ms_eager"], (d.get("value_sequential") or {}).get("frames_per_s"), d["roofline"]["dense_launch_ms"].get("tile_activity+fill"))
except Exception as ex:
    print("unreadable", ex)
PY
