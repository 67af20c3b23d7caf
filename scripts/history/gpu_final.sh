#!/bin/bash
# end-of-round artefacts: full GPU test suite, smoke, the three bench lines, the kernel traces of the two batch-1 lines, and the
# counters of the LDS-tiled stream-K conv kernel. Outputs under gpurun_out/final_*; the summaries are copied to profiles/ by hand.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 5 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/final_tests.log 2>&1
echo "tests exit $?"; tail -2 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench_2streams.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench_2streams.json')); r=d['roofline']; print('default', round(d['value'],1), round(d['ms_per_step'],4), d['stages_ms_eager'], 'roofline', round(r['avg_launch_ms']*1e3,1), round(r['frac'],3), 'cpu', d.get('cpu_baseline',{}).get('value'), 'host_io', d.get('host_io',{}).get('frames_per_s')); m=d['roofline_spmiddle']; print('spmiddle', m['frac'], m['mfma']['conv_ms'], m['mfma']['executed_tflops'])"
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/final_bench_1stream.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench_1stream.json')); print('1stream', round(d['value'],1), round(d['ms_per_step'],4), d.get('host_io',{}).get('frames_per_s'))"
timeout -k 5 400 python -u bench.py --stress --steps 30 --warmup 5 --cpu-frames 0 2> gpurun_out/final_stress.err | tail -1 > gpurun_out/final_bench_stress.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench_stress.json')); m=d['roofline_spmiddle'].pop('mfma'); print('stress', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms_eager'], d['roofline_spmiddle']['frac'], m['conv_ms'], m['executed_tflops'], m['executed_frac_of_f32_mfma_peak'])"
cd /tmp && export TMPDIR=/tmp
for s in 1 2; do
rm -rf $R/gpurun_out/prof_f$s
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f$s -o f$s -- python $R/bench.py --steps $((100*s)) --warmup $((10*s)) --cpu-frames 0 --streams $s --no-roofline --no-host-io > $R/gpurun_out/prof_f$s.log 2>&1
DB=$(find $R/gpurun_out/prof_f$s -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $((100*s)) 45 > $R/gpurun_out/final_trace_${s}streams.txt; head -3 $R/gpurun_out/final_trace_${s}streams.txt | cut -c1-150
find $R/gpurun_out/prof_f$s -name "*.db" -delete
done
cd $R
CSK_LAYERS=0 CSK_SKIP_DIRECT=1 PROBE=csk_probe.py PASSES=4 bash scripts/gpu_wino_pmc.sh 0 csk conv2d_sk_kernel > gpurun_out/final_csk_pmc.log 2>&1; tail -3 gpurun_out/csk_pmc_summary.txt
