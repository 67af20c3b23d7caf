#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k; mkdir -p $O
cd $R
timeout -k 5 1200 python -m pytest tests/test_bench_gpu.py tests/test_rccl_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log | cut -c1-600
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6k/bench_driver.json").read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print(round(d["value"], 1), {k: c[k] for k in c if k.startswith("parity")}, r["traffic"], r.get("traffic_times_algorithmic"), r.get("frac_chip_timed_region_from_counters"))
PY
