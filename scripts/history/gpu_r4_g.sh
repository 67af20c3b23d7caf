#!/bin/bash
# round 4, seventh GPU call: multi-tile sparse waves (tests + stress bench), ODIoU lane split (tests + train bench)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4g
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_conv_gpu.py tests/test_odiou_gpu.py tests/test_head_loss_gpu.py -q -m gpu -x > gpurun_out/r4g/tests.log 2>&1
echo "tests rc $?"; tail -8 gpurun_out/r4g/tests.log | cut -c1-300
SESSD_BENCH_VERBOSE=1 timeout 600 python bench.py --stress --steps 40 --warmup 5 --cpu-frames 8 > gpurun_out/r4g/bench_stress.json 2> gpurun_out/r4g/bench_stress.err
echo "stress rc $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4g/bench_stress.json"))
    print("stress", round(j["value"], 1), round(j["ms_per_step"], 4), "parity ok", (j.get("parity") or {}).get("ok"))
    print("  stages", j.get("stages_ms_eager"), "spm", round(j.get("roofline_spmiddle", {}).get("frac", 0), 4))
    m = j.get("roofline_spmiddle", {}).get("mfma", {})
    print("  sparse conv_ms", m.get("conv_ms"), "exec frac", m.get("executed_frac_of_f32_mfma_peak"))
    print("  tuning", j["config"]["tuning"]["sparse"])
    print("  per layer ms", [L["ms"] for L in m.get("layers", [])])
except Exception as e:
    print("unreadable", e)
PY
timeout 600 python scripts/train_step_bench.py --real-loss --steps 20 > gpurun_out/r4g/train_step.json 2> gpurun_out/r4g/train_step.err
echo "train bench rc $?"; cut -c1-200 gpurun_out/r4g/train_step.json | head -2; python -c "
import json; j=json.load(open('gpurun_out/r4g/train_step.json')); print({k:j[k] for k in ('ms_per_iter','capacity_form_eager_ms_per_iter','standin_loss_graph_ms_per_iter','loss')})"
