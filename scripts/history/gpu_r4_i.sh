#!/bin/bash
# round 4: bucketed hash (x-runs share a 32-byte bucket) -- every test that touches the hash, then the benches
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_voxelize_gpu.py tests/test_sparse_sites_gpu.py tests/test_sparse_conv_gpu.py tests/test_site_renumber_gpu.py tests/test_sparse_grad_gpu.py tests/test_abi.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 5 600 python bench.py --stress --no-train-step > $O/bench_stress.json 2>$O/bench_stress.err; echo "stress rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4i/bench_stress.json").read().strip().splitlines()[-1])
print("stress", d["value"], d["ms_per_step"], d["parity"].get("ok"), d["stages_ms_eager"])
PY
timeout -k 5 600 python bench.py --no-train-step > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4i/bench_default.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], d["parity"].get("ok"), d["stages_ms_eager"])
PY
