#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 600 python -m pytest tests/test_dense_active_gpu.py -x -q -m gpu 2>&1 | tail -3
python scripts/activity_probe.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: round(v,2) for k,v in d.items() if k.startswith('steps')})"
