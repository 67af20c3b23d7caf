R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r1w
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1w -o r1w -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 --no-roofline > $R/gpurun_out/prof_r1w.log 2>&1
tail -1 $R/gpurun_out/prof_r1w.log
DB=$(find $R/gpurun_out/prof_r1w -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 1 60 > $R/gpurun_out/prof_r1w_summary.txt
head -3 $R/gpurun_out/prof_r1w_summary.txt
