R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stress
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stress -o st -- python $R/bench.py --stress --steps 10 --warmup 2 --cpu-frames 0 --no-roofline > $R/gpurun_out/prof_stress.log 2>&1
tail -1 $R/gpurun_out/prof_stress.log | cut -c1-200
DB=$(find $R/gpurun_out/prof_stress -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 10 40 > $R/gpurun_out/prof_stress_summary.txt; head -30 $R/gpurun_out/prof_stress_summary.txt | cut -c1-90,100-175
