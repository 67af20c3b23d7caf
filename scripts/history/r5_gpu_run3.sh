#!/bin/bash
# round 5, GPU call 3: the bench line with the pre-trained train_step leg; activity kernel back on 16 waves; trained-weights line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py tests/test_train_gpu.py tests/test_trained_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver rc $?"
timeout -k 5 600 python bench.py --streams 1 --no-train-step --no-host-io > $O/bench_1stream.json 2>$O/bench_1stream.err; echo "1stream rc $?"
python - <<'PY'
import json
for n in ("driver", "1stream"):
    try:
        d = json.loads(open("gpurun_out/r5c/bench_%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], "frac", round(r["frac"], 3), {k: round(v * 1e3, 1) for k, v in r["dense_launch_ms"].items()}, d["stages_ms_eager"],
              (d.get("value_sequential") or {}).get("frames_per_s"), d.get("train_step"))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
