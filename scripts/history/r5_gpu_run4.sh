#!/bin/bash
# round 5: the trained-weights parity record on a second seed and on a longer run (is engine == oracle a property of one model?)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
timeout -k 5 900 python tests/trained_parity.py --iterations 2000 --scenes 400 --heldout 200 --seed 1 --workers 12 --out $O/trained_parity_seed1.json > $O/seed1.log 2>&1; echo "seed1 rc $?"; tail -22 $O/seed1.log | head -30
timeout -k 5 1200 python tests/trained_parity.py --iterations 5000 --scenes 800 --heldout 200 --seed 2 --workers 12 --out $O/trained_parity_seed2_5000.json > $O/seed2.log 2>&1; echo "seed2 rc $?"; tail -22 $O/seed2.log | head -30
