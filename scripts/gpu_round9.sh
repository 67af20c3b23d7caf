timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export SESSD_BENCH_VERBOSE=1
timeout 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --streams 1 2>&1 | grep "timed region\|roofline kernel\|autotuned" | cut -c1-900
timeout 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline 2>&1 | grep "timed region"
