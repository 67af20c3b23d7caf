timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pf --output-format csv -- python $R/scripts/conv_occupancy_probe2.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pw --output-format csv -- python $R/scripts/conv_occupancy_probe2.py > $R/gpurun_out/pmc_write.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
