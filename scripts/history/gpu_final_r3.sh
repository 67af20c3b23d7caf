#!/bin/bash
# end-of-round artefacts (round 3): full GPU test suite, smoke, the bench lines (default two frames in flight = the driver's
# command, one frame at a time, --stress), kernel traces of the two batch-1 lines, the training iteration eager vs captured with
# its kernel trace, the data-path bench. Outputs under gpurun_out/final_*; summaries are copied to profiles/r3_* by hand.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
if [ -z "${NOTESTS:-}" ]; then
timeout -k 5 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/final_tests.log 2>&1
echo "tests exit $?"; tail -2 gpurun_out/final_tests.log
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py 2> gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench_2streams.json; echo "bench rc $?"; python -c "
import json; d=json.load(open('gpurun_out/final_bench_2streams.json')); r=d['roofline']; p=d['parity']; print('default', round(d['value'],1), round(d['ms_per_step'],4), 'seq', d['value_sequential']['frames_per_s'], d['stages_ms_eager'], 'roofline', round(r['avg_launch_ms']*1e3,1), round(r['frac'],3), 'cpu', d.get('cpu_baseline',{}).get('value'), 'host_io', d.get('host_io',{}).get('frames_per_s'), d.get('host_io',{}).get('latency_mode_frames_per_s'), 'parity', p['frames'], p['identical'], p['flipped_near_threshold'], p['ok']); m=d['roofline_spmiddle']; print('spmiddle', m['ms'], m['frac'], m['mfma']['conv_ms'], m['mfma']['executed_tflops'], m['mfma']['useful_row_fraction'])"
timeout -k 5 200 python -u bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final_bench_driver_cmd.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench_driver_cmd.json')); print('driver command (--steps 20 --warmup 5)', round(d['value'],1), d['parity']['ok'])"
timeout -k 5 200 python -u bench.py --streams 1 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/final_bench_1stream.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench_1stream.json')); print('1stream', round(d['value'],1), round(d['ms_per_step'],4), d['stages_ms_eager'])"
timeout -k 5 400 python -u bench.py --stress --steps 30 --warmup 5 --cpu-frames 0 2> gpurun_out/final_stress.err | tail -1 > gpurun_out/final_bench_stress.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench_stress.json')); m=d['roofline_spmiddle'].pop('mfma'); print('stress', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms_eager'], d['roofline_spmiddle']['frac'], m['conv_ms'], m['executed_tflops'], m['executed_frac_of_f32_mfma_peak'], m['useful_row_fraction'])"
timeout -k 5 300 python scripts/train_step_bench.py --steps 10 --graph 2> gpurun_out/final_train.err | tail -1 > gpurun_out/final_train_step.json; cut -c1-700 gpurun_out/final_train_step.json
timeout -k 5 300 python scripts/datapath_bench.py 2>/dev/null | tail -1 > gpurun_out/final_datapath.json; cut -c1-600 gpurun_out/final_datapath.json
cd /tmp && export TMPDIR=/tmp
for s in 1 2; do
rm -rf $R/gpurun_out/prof_f$s
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f$s -o f$s -- python $R/bench.py --steps $((100*s)) --warmup $((10*s)) --cpu-frames 0 --streams $s --no-roofline --no-host-io --no-sequential > $R/gpurun_out/prof_f$s.log 2>&1
DB=$(find $R/gpurun_out/prof_f$s -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $((100*s)) 50 > $R/gpurun_out/final_trace_${s}streams.txt; head -3 $R/gpurun_out/final_trace_${s}streams.txt | cut -c1-150
rm -rf $R/gpurun_out/prof_f$s
done
rm -rf $R/gpurun_out/prof_train
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o tr -- python $R/scripts/train_step_bench.py --steps 10 --graph > $R/gpurun_out/prof_train.log 2>&1
DB=$(find $R/gpurun_out/prof_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 0 70 > $R/gpurun_out/final_train_trace.txt; head -3 $R/gpurun_out/final_train_trace.txt | cut -c1-150
rm -rf $R/gpurun_out/prof_train
