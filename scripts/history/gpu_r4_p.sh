#!/bin/bash
# round 4: activity kernel with wave-per-word list emission -- tests, trace of the sequential frame, bench
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py -x -q -m gpu > $O/tests_active.log 2>&1; echo "active tests rc $?"; tail -5 $O/tests_active.log
timeout -k 5 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "active_tiles" > $O/tests_pipe.log 2>&1; echo "pipeline tests rc $?"; tail -5 $O/tests_pipe.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p -o t -- python $R/bench.py --steps 100 --warmup 10 --streams 1 --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p.log 2>&1
echo "rc $?"
DB=$(find $O/p -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 100 60 > $O/trace_1stream.txt; head -24 $O/trace_1stream.txt | cut -c1-175
rm -rf $O/p
cd $R
timeout -k 5 300 python bench.py --no-train-step --no-host-io --no-roofline --cpu-frames 8 > $O/bench.json 2>$O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4p/bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d.get("value_sequential",{}).get("frames_per_s"), d["config"]["tuning"].get("active_tiles"))
PY
