R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -x -q > gpurun_out/all.log 2>&1; tail -3 gpurun_out/all.log | cut -c1-160
timeout -k 5 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 100 python scripts/odiou_bench.py 2>/dev/null | tail -1 | tee gpurun_out/odiou_bench.json
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; python -c "
import json; j=json.load(open('gpurun_out/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'], j['host_io']['frames_per_s'], j['cpu_baseline']['value'])"
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/bench_1stream.json; python -c "
import json; j=json.load(open('gpurun_out/bench_1stream.json')); print(j['value'], j['ms_per_step'], j['stages_ms_eager'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_fin
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fin -o fin -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 --no-roofline > $R/gpurun_out/prof_fin.log 2>&1
DB=$(find $R/gpurun_out/prof_fin -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 100 40 > $R/gpurun_out/prof_fin_summary.txt; head -3 $R/gpurun_out/prof_fin_summary.txt | cut -c1-150
