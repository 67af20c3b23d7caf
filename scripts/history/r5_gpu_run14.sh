#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?"; grep -E "share of|kind |passed|failed|Error" $O/tests.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { timeout -k 5 300 python bench.py --steps 800 --warmup 80 --cpu-frames 8 --no-host-io --no-sequential --no-train-step $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"; }
run def_a ""
run def_b ""
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5v/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["parity"]["frames"], round(r["dense_launch_ms"]["tile_activity+fill"] * 1e3, 1), d["stages_ms_eager"])
PY
