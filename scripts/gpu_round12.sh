R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export SESSD_BENCH_VERBOSE=1
cd $R
timeout -k 5 300 python -u bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; tail -c 600 gpurun_out/bench_default.json; echo
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/bench_1stream.json; tail -c 300 gpurun_out/bench_1stream.json; echo
timeout -k 5 400 python -u bench.py --stress --steps 30 --warmup 5 --cpu-frames 0 2> gpurun_out/bench_stress.err | tail -1 > gpurun_out/bench_stress.json; tail -c 900 gpurun_out/bench_stress.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r1x
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1x -o r1x -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 --no-roofline > $R/gpurun_out/prof_r1x.log 2>&1
DB=$(find $R/gpurun_out/prof_r1x -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 100 40 > $R/gpurun_out/prof_r1x_summary.txt; head -4 $R/gpurun_out/prof_r1x_summary.txt | cut -c1-150
for C in FETCH_SIZE WRITE_SIZE; do
timeout -k 5 90 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc8_$C -o p --output-format csv -- python $R/scripts/wino_probe.py 20 > $R/gpurun_out/pmc8_$C.log 2>&1
f=$(find $R/gpurun_out/pmc8_$C -name "*counter_collection.csv" | head -1); python $R/scripts/pmc_summary.py $f winograd
done
