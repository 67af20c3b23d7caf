set -x
mkdir -p gpurun_out
timeout 120 python scripts/debug_module_path.py 2>&1 | tail -8
timeout 300 python bench.py --steps 300 --warmup 30 2>&1 | tail -3 | tee gpurun_out/bench_graph.json
timeout 200 python bench.py --steps 100 --warmup 10 --eager --cpu-frames 0 --no-roofline 2>&1 | tail -1 | tee gpurun_out/bench_eager.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o r1 -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 > $R/gpurun_out/prof_r1.log 2>&1
tail -2 $R/gpurun_out/prof_r1.log
find $R/gpurun_out/prof_r1 -type f | head -20
