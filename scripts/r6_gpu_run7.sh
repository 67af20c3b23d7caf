#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g; mkdir -p $O $R/build
cd $R
timeout -k 5 600 python -m pytest tests/test_cu_mask_gpu.py -q -m gpu -s > $O/cu_mask.log 2>&1; echo "cu mask rc $?"; grep -v "^$" $O/cu_mask.log | tail -25 | cut -c1-400
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --save-weights build/r6_student.pt > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver rc $?"
timeout -k 5 600 python bench.py --weights build/r6_student.pt --no-train-step --no-host-io > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"
python - <<'PY'
import json
for n in ("driver", "default"):
    d = json.loads(open("gpurun_out/r6g/bench_%s.json" % n).read().strip().splitlines()[-1])
    c, r = d["config"], d["roofline"]
    print(n, round(d["value"], 1), c["parity_ok"], c["parity_matched"], c["parity_frames"], c["parity_rule"], r["frac"], r["frac_of_cu_set_peak"], r["frac_list_launches_of_cu_set_peak"], r["frac_chip_timed_region"], r.get("frac_chip_timed_region_from_counters"), r.get("traffic_times_algorithmic"),
          {k: round(v * 1e3, 1) for k, v in r["dense_launch_ms"].items()}, d["tuning"]["active_tiles"])
PY
