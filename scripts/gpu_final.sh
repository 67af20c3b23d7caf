timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_final.json; cut -c1-400 gpurun_out/bench_final.json
timeout 300 python bench.py --streams 1 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/bench_final_s1.json; cut -c1-300 gpurun_out/bench_final_s1.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | cut -c1-200
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 > $R/gpurun_out/prof_final.log 2>&1
tail -1 $R/gpurun_out/prof_final.log | cut -c1-200
