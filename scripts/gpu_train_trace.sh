#!/bin/bash
# training iteration: chosen tests ($TESTS), eager vs captured timing, and the kernel table (rocprofv3 --kernel-trace) of the
# same command with up to 120 rows. Outputs under gpurun_out/train_<tag>_*.
set -u
TAG=${1:-t}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
if [ -n "${TESTS:-}" ]; then
  timeout -k 5 ${TEST_TIMEOUT:-600} python -m pytest $TESTS -m gpu -q -x --timeout 300 > gpurun_out/train_${TAG}_tests.log 2>&1
  echo "tests exit $?"; tail -${TAIL:-4} gpurun_out/train_${TAG}_tests.log
fi
timeout -k 5 300 python scripts/train_step_bench.py --steps 10 --graph 2> gpurun_out/train_${TAG}.err | tail -1 > gpurun_out/train_${TAG}_step.json; cut -c1-700 gpurun_out/train_${TAG}_step.json
if [ -z "${NOTRACE:-}" ]; then
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o tr -- python $R/scripts/train_step_bench.py --steps 10 --graph > $R/gpurun_out/prof_train.log 2>&1
DB=$(find $R/gpurun_out/prof_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 0 120 > $R/gpurun_out/train_${TAG}_trace.txt; head -3 $R/gpurun_out/train_${TAG}_trace.txt | cut -c1-150
rm -rf $R/gpurun_out/prof_train
fi
