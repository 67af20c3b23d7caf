timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
export SESSD_BENCH_VERBOSE=1
timeout 200 python -u bench.py --steps 200 --warmup 20 2>&1 | tail -12 | tee gpurun_out/bench_r1b.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1b -o r1b -- python $R/bench.py --steps 60 --warmup 10 --cpu-frames 0 --eager > $R/gpurun_out/prof_r1b.log 2>&1
tail -2 $R/gpurun_out/prof_r1b.log
