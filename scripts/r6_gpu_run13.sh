#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6l
timeout -k 5 300 python scripts/r6_fill_probe.py > gpurun_out/r6l/fill_probe.json 2>gpurun_out/r6l/fill_probe.err; echo "rc $?"; cat gpurun_out/r6l/fill_probe.json | head -80; tail -3 gpurun_out/r6l/fill_probe.err
