#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python scripts/sparse_layer_probe.py --layer 0 1 2 3 5 6 9 10 13 2>&1 | grep "layer\|sites"
python scripts/sparse_layer_probe.py --stress --layer 0 1 2 3 5 6 9 10 13 --reps 10 2>&1 | grep "layer\|sites"
for sp in 1 2 4; do python scripts/sparse_layer_probe.py --stress --layer 6 10 --split $sp --depth 3 --reps 10 2>&1 | grep "^layer" | sed "s/^/split $sp: /"; done
