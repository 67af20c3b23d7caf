#!/bin/bash
# round 4: fill only where the reader can reach -- tests (lists, pipeline), bench with its parity gate, trace of the sequential frame
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
timeout -k 5 300 python -m pytest tests/test_dense_active_gpu.py -x -q -m gpu > $O/tests_active.log 2>&1; echo "active tests rc $?"; tail -4 $O/tests_active.log
timeout -k 5 300 python -m pytest tests/test_pipeline_gpu.py tests/test_runner_gpu.py -x -q -m gpu > $O/tests_pipe.log 2>&1; echo "pipeline tests rc $?"; tail -4 $O/tests_pipe.log
timeout -k 5 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-train-step > $O/bench_driver.json 2>$O/bench_driver.err; echo "bench rc $?"
timeout -k 5 300 python bench.py --stress --no-train-step --no-host-io > $O/bench_stress.json 2>$O/bench_stress.err; echo "stress rc $?"
python - <<'PY'
import json
for n in ("driver","stress"):
    try:
        d=json.loads(open("gpurun_out/r4s/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"), d["parity"].get("frames"), d["stages_ms_eager"], (d.get("value_sequential") or {}).get("frames_per_s"), d["roofline"]["dense_launch_ms"].get("tile_activity+fill"))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
