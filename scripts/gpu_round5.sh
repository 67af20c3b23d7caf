timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
export SESSD_BENCH_VERBOSE=1
timeout 200 python -u bench.py --steps 300 --warmup 30 2>&1 | grep -v "cpu frame\|autotuned" | tail -4 | cut -c1-2200
