#!/bin/bash
# round 6: the trained-parity records again on the FINAL engine (all ten active-tile layers incl. conv_0 / conv_1 over their list): seeds 0 and 2
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
timeout -k 5 900 python tests/trained_parity.py --iterations 2000 --scenes 400 --heldout 200 --seed 0 --out $O/trained_parity_seed0.json > $O/seed0.log 2>&1; echo "seed 0 rc $?"
timeout -k 5 1200 python tests/trained_parity.py --iterations 5000 --scenes 800 --heldout 200 --seed 2 --out $O/trained_parity_seed2_5000it.json > $O/seed2.log 2>&1; echo "seed 2 rc $?"
python - <<'PY'
import json
for n in ("seed0", "seed2_5000it"):
    d = json.load(open("gpurun_out/r6j/trained_parity_%s.json" % n))
    print(n, {k: v for k, v in d["engine_vs_oracle_strict"].items() if k != "mismatch"}, d["car_3d_ap_0p7_moderate"], d["training"]["sparse_overflow_flag"], d["training"]["overflow_flags"],
          d["training_checks"]["moving_average_first_last"], d["active_tile_layers_of_the_engine"], round(d["training"]["samples_per_s"], 1))
PY
