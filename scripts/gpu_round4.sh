timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 100 python scripts/conv_occupancy_probe2.py 2>&1 | tail -4
export SESSD_BENCH_VERBOSE=1
timeout 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 2>&1 | grep -v "cpu frame" | tail -6 | cut -c1-1200
timeout 200 python -u bench.py --steps 300 --warmup 30 --streams 2 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | cut -c1-400
