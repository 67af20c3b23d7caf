#!/bin/bash
# round 4, fifth GPU call: offset-pattern tiles (tests, benches)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_sites_gpu.py tests/test_sparse_conv_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x -k "chain_sites or offset_pattern or engine_vs_oracle or every_tuning" > gpurun_out/r4e/tests.log 2>&1
echo "tests rc $?"; tail -14 gpurun_out/r4e/tests.log | cut -c1-300
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py --stress --steps 40 --warmup 5 --cpu-frames 8 > gpurun_out/r4e/bench_stress.json 2> gpurun_out/r4e/bench_stress.err
echo "stress rc $?"
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py --no-train-step > gpurun_out/r4e/bench_default.json 2> gpurun_out/r4e/bench_default.err
echo "default rc $?"
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py --no-train-step --streams 1 --cpu-frames 0 --no-host-io > gpurun_out/r4e/bench_1stream.json 2> gpurun_out/r4e/bench_1stream.err
echo "1stream rc $?"
python - <<'PY'
import json
for f in ("bench_stress", "bench_default", "bench_1stream"):
    try:
        j = json.load(open("gpurun_out/r4e/%s.json" % f))
        print(f, round(j["value"], 1), round(j["ms_per_step"], 4), "parity ok", (j.get("parity") or {}).get("ok"), "seq", (j.get("value_sequential") or {}).get("frames_per_s"))
        print("  stages", j.get("stages_ms_eager"), "roofline", round(j.get("roofline", {}).get("frac", 0), 4), "spm", round(j.get("roofline_spmiddle", {}).get("frac", 0), 4))
        m = j.get("roofline_spmiddle", {}).get("mfma", {})
        print("  sparse conv_ms", m.get("conv_ms"), "useful", m.get("useful_row_fraction"), "exec frac", m.get("executed_frac_of_f32_mfma_peak"))
        print("  sorted", j["config"]["tuning"].get("sparse_offset_pattern_tiles"))
        print("  per layer ms", [L["ms"] for L in m.get("layers", [])])
    except Exception as e:
        print(f, "unreadable", e)
PY
