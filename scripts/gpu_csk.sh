#!/bin/bash
# tests of the LDS-tiled stream-K conv kernel + its timing probe (+ the f32 MFMA rate microbenchmark with UBENCH=1)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 400 python -m pytest tests/test_dense_conv_gpu.py -m gpu -q -x --timeout 300 -k "lds_stream_k" > gpurun_out/csk_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/csk_tests.log
timeout -k 5 200 python scripts/csk_probe.py ${CSK_WGS:-0} > gpurun_out/csk_probe.log 2>&1
echo "probe exit $?"; grep "cfg 30" gpurun_out/csk_probe.log | tail -40
if [ "${UBENCH:-0}" = "1" ]; then
  hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 60 /tmp/mfma_peak > gpurun_out/mfma_peak.log 2>&1
  grep -i "blocks=256\|blocks=1024" gpurun_out/mfma_peak.log
fi
