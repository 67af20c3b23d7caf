#!/bin/bash
# round 4, first GPU call: the new parity / pipeline tests, then the stress bench (with its parity gate) and the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_runner_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r4a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r4a/tests.log
tail -5 gpurun_out/r4a/tests.log
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py --stress --steps 40 --warmup 5 > gpurun_out/r4a/bench_stress.json 2> gpurun_out/r4a/bench_stress.err
echo "stress rc $?"
tail -c 1500 gpurun_out/r4a/bench_stress.err
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
echo "default rc $?"
python - <<'PY'
import json
for f in ("bench_stress", "bench_default"):
    try:
        j = json.load(open("gpurun_out/r4a/%s.json" % f))
        print(f, j["value"], j["ms_per_step"], {k: v for k, v in (j.get("parity") or {}).items() if k not in ("rule",)})
        print("  stages", j.get("stages_ms_eager"), "roofline", j.get("roofline", {}).get("frac"), "spm", j.get("roofline_spmiddle", {}).get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
