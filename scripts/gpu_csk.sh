#!/bin/bash
# tests of the LDS-tiled stream-K conv kernel and of the paired-class transposed conv + their timing probe
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 400 python -m pytest tests/test_dense_conv_gpu.py -m gpu -q -x --timeout 300 -k "lds_stream_k or paired" > gpurun_out/csk_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/csk_tests.log
timeout -k 5 200 python scripts/csk_probe.py ${CSK_WGS:-0} > gpurun_out/csk_probe.log 2>&1
echo "probe exit $?"; grep "cfg" gpurun_out/csk_probe.log | tail -40
