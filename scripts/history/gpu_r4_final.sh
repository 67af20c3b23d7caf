#!/bin/bash
# round 4: full validation -- the whole GPU suite, smoke(), the bench lines that go into profiles/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final; mkdir -p $O
cd $R
timeout -k 5 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver-line rc $?"
timeout -k 5 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"
timeout -k 5 600 python bench.py --streams 1 --no-train-step > $O/bench_1stream.json 2>$O/bench_1stream.err; echo "1stream rc $?"
timeout -k 5 600 python bench.py --stress > $O/bench_stress.json 2>$O/bench_stress.err; echo "stress rc $?"
python - <<'PY'
import json
for n in ("driver","default","1stream","stress"):
    try:
        d=json.loads(open("gpurun_out/r4final/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["roofline"]["frac"], d.get("stages_ms_eager"), (d.get("train_step") or {}).get("ms_per_iter"))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
