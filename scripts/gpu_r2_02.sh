#!/bin/bash
# Round 2, GPU call 2: scalar-control sparse conv kernel: parity + timing + counters.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 500 python -m pytest tests/test_sparse_sites_gpu.py tests/test_sparse_conv_gpu.py tests/test_site_renumber_gpu.py tests/test_pipeline_gpu.py -q -x --timeout 200 > gpurun_out/r2_02_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/r2_02_tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in b1 stress; do
  flag=""; [ $cfg = stress ] && flag="--stress"
  timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sp2_${cfg}_trace -o t -- python $R/scripts/sparse_probe.py $flag --frames 6 > $R/gpurun_out/sp2_${cfg}_trace.log 2>&1
  grep "sites\|stages\|tuning" $R/gpurun_out/sp2_${cfg}_trace.log
  DB=$(find $R/gpurun_out/sp2_${cfg}_trace -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB 4 40 > $R/gpurun_out/sp2_${cfg}_trace_summary.txt
  grep "sparse_conv\|rulebook\|down_insert\|kernel time" $R/gpurun_out/sp2_${cfg}_trace_summary.txt | cut -c1-60,100-170
  find $R/gpurun_out/sp2_${cfg}_trace -name "*.db" -delete
done
cd $R
timeout -k 5 300 python bench.py --streams 1 --cpu-frames 0 > gpurun_out/r2_02_bench_1stream.json 2> gpurun_out/r2_02_bench.err; tail -c 1500 gpurun_out/r2_02_bench_1stream.json
