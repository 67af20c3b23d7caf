timeout -k 5 600 python -m pytest tests/test_sparse_conv_gpu.py tests/test_pipeline_gpu.py tests/test_sparse_grad_gpu.py -m gpu -x -q 2>&1 | tail -2
export SESSD_BENCH_VERBOSE=1
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --streams 1 --no-roofline 2>&1 | grep "timed region"
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline 2>&1 | grep "timed region"
timeout -k 5 300 python -u bench.py --stress --steps 30 --warmup 5 --cpu-frames 0 2>&1 | grep "timed region"
