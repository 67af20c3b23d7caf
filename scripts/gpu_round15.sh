export SESSD_BENCH_VERBOSE=1
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --streams 1 --no-roofline 2>&1 | grep "timed region\|autotuned" | sed 's/.*deconv_0/deconv_0/' | cut -c1-200
timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline 2>&1 | grep "timed region"
timeout -k 5 300 python -m pytest tests/test_dense_conv_gpu.py -m gpu -x -q 2>&1 | tail -1
