timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export SESSD_BENCH_VERBOSE=1
timeout 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-autotune 2>&1 | grep "timed region\|roofline kernel"
timeout 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --streams 2 --no-roofline --no-autotune 2>&1 | grep "timed region"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1f -o r1f -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --eager --no-roofline --no-autotune > $R/gpurun_out/prof_r1f.log 2>&1
