#!/bin/bash
# round 5: the CU-set configuration as bench.py's default -- its test, the driver's command, the legs, repeats against the old default
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "cu_sets" > $O/test_cu_sets.log 2>&1; echo "test rc $?"; tail -3 $O/test_cu_sets.log
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver rc $?"; tail -2 $O/bench_driver.err | cut -c1-300
timeout -k 5 600 python bench.py --no-train-step > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"
timeout -k 5 600 python bench.py --no-train-step --streams 2 --cu-split none > $O/bench_r4_config.json 2>$O/bench_r4_config.err; echo "r4 config rc $?"
GPU_MAX_HW_QUEUES=8 timeout -k 5 600 python bench.py --no-train-step --no-host-io --no-roofline --no-sequential --cpu-frames 8 > $O/bench_hwq8.json 2>$O/bench_hwq8.err; echo "hwq8 rc $?"
GPU_MAX_HW_QUEUES=8 timeout -k 5 600 python bench.py --no-train-step --no-host-io --no-roofline --no-sequential --cpu-frames 8 --streams 2 --cu-split none > $O/bench_hwq8_r4.json 2>$O/bench_hwq8_r4.err; echo "hwq8 r4 rc $?"
python - <<'PY'
import json
for n in ("driver", "default", "r4_config", "hwq8", "hwq8_r4"):
    try:
        d = json.loads(open("gpurun_out/r5l/bench_%s.json" % n).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(n, round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], d["config"]["frames_in_flight"], d["config"].get("cu_sets"),
              round(d["config"].get("ms_latency_per_frame_in_flight", 0), 3), "frac", r.get("frac"), r.get("frac_of_whole_chip_peak"), r.get("frac_full_map_launches"), r.get("frac_list_launches"),
              (d.get("roofline_whole_chip_engine") or {}).get("frac"), d.get("stages_ms_eager"), d.get("stages_ms_eager_whole_chip_engine"),
              (d.get("value_sequential") or {}).get("frames_per_s"), {k: (d.get("host_io") or {}).get(k) for k in ("frames_per_s", "latency_mode_frames_per_s")},
              {k: round(v * 1e3, 1) for k, v in (r.get("dense_launch_ms") or {}).items()})
    except Exception as ex:
        print(n, "unreadable", ex)
PY
