#!/bin/bash
# round 4: direct kernels over tile lists (1x1 layers, the pair of transposed convs) -- tests, bench
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4r; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py -x -q -m gpu > $O/tests_active.log 2>&1; echo "active tests rc $?"; tail -15 $O/tests_active.log
timeout -k 5 600 python -m pytest tests/test_pipeline_gpu.py tests/test_dense_conv_gpu.py -x -q -m gpu -k "active_tiles or whatever_autotune or stress_autotuned or deconv or conv2d_1x1 or pair" > $O/tests_pipe.log 2>&1; echo "pipeline tests rc $?"; tail -8 $O/tests_pipe.log
timeout -k 5 600 python bench.py --no-train-step --no-host-io > $O/bench_on.json 2>$O/bench_on.err; echo "bench rc $?"; tail -3 $O/bench_on.err
timeout -k 5 600 python bench.py --stress --no-train-step --no-host-io > $O/stress_on.json 2>$O/stress_on.err; echo "stress rc $?"
python - <<'PY'
import json
for n in ("bench_on","stress_on"):
    try:
        d=json.loads(open("gpurun_out/r4r/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"), round(d["roofline"]["frac"],3), d["roofline"].get("frac_full_map_launches"), d["stages_ms_eager"], d.get("value_sequential",{}).get("frames_per_s"), d["config"]["tuning"].get("active_tiles"), d["roofline"].get("active_tile_fraction"), d["roofline"]["dense_launch_ms"])
    except Exception as ex:
        print(n, "unreadable", ex)
PY
