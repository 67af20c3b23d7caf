timeout 60 python -u scripts/debug_graph.py sync 2>&1 | tail -22
echo ----
timeout 60 python -u scripts/debug_graph.py nosync 2>&1 | tail -8
echo ---- profile eager
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o r1 -- python $R/bench.py --steps 60 --warmup 10 --cpu-frames 0 --eager --no-roofline > $R/gpurun_out/prof_r1.log 2>&1
tail -2 $R/gpurun_out/prof_r1.log
find $R/gpurun_out/prof_r1 -type f | head
