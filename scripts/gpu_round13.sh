timeout -k 5 300 python -m pytest tests/test_sparse_conv_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3
export SESSD_BENCH_VERBOSE=1
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --streams 1 2>&1 | grep "timed region\|autotuned\|stages" | cut -c1-420
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline 2>&1 | grep "timed region"
