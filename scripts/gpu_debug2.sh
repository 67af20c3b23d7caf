export SESSD_BENCH_VERBOSE=1
nproc; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"
timeout 100 python -u bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-roofline 2>&1 | tail -12
echo ---- eager+cpu
timeout 150 python -u bench.py --steps 20 --warmup 5 --eager --cpu-frames 2 --no-roofline 2>&1 | tail -12
