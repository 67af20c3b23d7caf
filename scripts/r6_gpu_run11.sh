#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
timeout -k 5 900 python tests/trained_parity.py --iterations 2000 --scenes 400 --heldout 200 --seed 1 --out $O/trained_parity_seed1.json > $O/seed1.log 2>&1; echo "seed 1 rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6j/trained_parity_seed1.json"))
print({k: v for k, v in d["engine_vs_oracle_strict"].items() if k != "mismatch"}, d["car_3d_ap_0p7_moderate"], d["training"]["sparse_overflow_flag"], len(d["active_tile_layers_of_the_engine"]))
PY
