timeout 300 python -m pytest tests/test_dense_conv_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 100 python scripts/conv_occupancy_probe2.py 2>&1 | tail -4
timeout 100 python scripts/conv_microbench.py 20 3,4 2>&1 | tail -10
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch2 -o pf --output-format csv -- python $R/scripts/conv_occupancy_probe2.py > $R/gpurun_out/pmc_fetch2.log 2>&1
