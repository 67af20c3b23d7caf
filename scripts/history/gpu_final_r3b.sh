#!/bin/bash
# end of round 3 (second half): the whole GPU suite, smoke(), the default bench line, the training iteration + its kernel table
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/final2_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/final2_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/final2_bench.err | tail -1 > gpurun_out/final2_bench.json; echo "bench exit $?"; cut -c1-400 gpurun_out/final2_bench.json
NOTRACE= bash scripts/gpu_train_trace.sh final2 | tail -3 | cut -c1-500
