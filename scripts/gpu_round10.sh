timeout -k 5 100 python scripts/conv_occupancy_probe2.py
timeout -k 5 900 python -m pytest tests -m gpu -x -q -s -k "stress" 2>&1 | grep -v "^$" | tail -6
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "not stress" 2>&1 | tail -3
export SESSD_BENCH_VERBOSE=1
for S in 1 2 3; do
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline --streams $S 2>&1 | grep "timed region\|autotuned" | cut -c1-700
done
