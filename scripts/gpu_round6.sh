timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1d -o r1d -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --eager --no-roofline --no-autotune > $R/gpurun_out/prof_r1d.log 2>&1
tail -1 $R/gpurun_out/prof_r1d.log | cut -c1-300
