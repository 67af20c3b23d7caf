timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
export SESSD_BENCH_VERBOSE=1
timeout 200 python -u bench.py --steps 300 --warmup 30 2>&1 | grep -v "cpu frame" | tail -8 | tee gpurun_out/bench_r1c.log
timeout 200 python -u bench.py --steps 300 --warmup 30 --streams 2 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | tee gpurun_out/bench_r1c_s2.log
timeout 200 python -u bench.py --steps 300 --warmup 30 --streams 3 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | tee gpurun_out/bench_r1c_s3.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1c -o r1c -- python $R/bench.py --steps 60 --warmup 10 --cpu-frames 0 --eager --no-roofline > $R/gpurun_out/prof_r1c.log 2>&1
tail -1 $R/gpurun_out/prof_r1c.log
