R=$GRAFT_REPO_ROOT
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; python -c "
import json; j=json.load(open('gpurun_out/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'], j.get('host_io'), j['cpu_baseline']['value'])"
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/bench_1stream.json; python -c "
import json; j=json.load(open('gpurun_out/bench_1stream.json')); print(j['value'], j['ms_per_step'], j.get('host_io'))"
timeout -k 5 200 python -u bench.py --batch 4 --streams 1 --cpu-frames 0 --steps 100 2>/dev/null | tail -1 > gpurun_out/bench_batch4.json; python -c "
import json; j=json.load(open('gpurun_out/bench_batch4.json')); print(j['value'], j['ms_per_step'], j['stages_ms_eager'])"
