#!/bin/bash
# round 4, fourth GPU call: batched voxelizer, Winograd tile_cfg 24, SyncBN two-rank test, benches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4d
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_voxelize_gpu.py tests/test_dense_conv_gpu.py -q -m gpu -x -k "four_launches or winograd_stream_k" > gpurun_out/r4d/tests_a.log 2>&1
echo "tests A rc $?"; tail -12 gpurun_out/r4d/tests_a.log
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "sync_bn" > gpurun_out/r4d/tests_b.log 2>&1
echo "tests B rc $?"; tail -12 gpurun_out/r4d/tests_b.log
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -x -k "mixed_cap or stress_config or engine_vs_oracle" > gpurun_out/r4d/tests_c.log 2>&1
echo "tests C rc $?"; tail -6 gpurun_out/r4d/tests_c.log
timeout 300 python scripts/wino_rk_probe.py > gpurun_out/r4d/wino_rk_probe.json 2> gpurun_out/r4d/wino_rk_probe.err
echo "probe rc $?"; cat gpurun_out/r4d/wino_rk_probe.json; tail -3 gpurun_out/r4d/wino_rk_probe.err
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py --stress --steps 40 --warmup 5 --cpu-frames 8 > gpurun_out/r4d/bench_stress.json 2> gpurun_out/r4d/bench_stress.err
echo "stress rc $?"
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py > gpurun_out/r4d/bench_default.json 2> gpurun_out/r4d/bench_default.err
echo "default rc $?"; grep "dense tile_cfg\|parity" gpurun_out/r4d/bench_default.err | cut -c1-400
python - <<'PY'
import json
for f in ("bench_stress", "bench_default"):
    try:
        j = json.load(open("gpurun_out/r4d/%s.json" % f))
        print(f, round(j["value"], 1), round(j["ms_per_step"], 4), "parity ok", (j.get("parity") or {}).get("ok"), "seq", (j.get("value_sequential") or {}).get("frames_per_s"))
        print("  stages", j.get("stages_ms_eager"), "roofline", round(j.get("roofline", {}).get("frac", 0), 4), j.get("roofline", {}).get("avg_launch_ms"), "spm", round(j.get("roofline_spmiddle", {}).get("frac", 0), 4))
        print("  dense", j.get("roofline", {}).get("dense_launch_ms"))
        print("  train_step", {k: v for k, v in (j.get("train_step") or {}).items() if k in ("ms_per_iter", "samples_per_s", "error", "loss")})
    except Exception as e:
        print(f, "unreadable", e)
PY
