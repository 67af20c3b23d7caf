#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 600 python -m pytest "$@" -q -x --timeout 600 2>&1 | grep -v "^$" | tail -25 | cut -c1-300
