#!/bin/bash
# round-3 iteration loop: a chosen set of GPU tests ($TESTS, pytest arguments), the sequential bench line and the kernel table
# of its timed region. Outputs under gpurun_out/r3_<tag>_*.
set -u
TAG=${1:-chk}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
if [ -n "${TESTS:-}" ]; then
  timeout -k 5 ${TEST_TIMEOUT:-500} python -m pytest $TESTS -m gpu -q -x --timeout 300 > gpurun_out/r3_${TAG}_tests.log 2>&1
  echo "tests exit $?"; tail -${TAIL:-6} gpurun_out/r3_${TAG}_tests.log
fi
if [ -z "${NOBENCH:-}" ]; then
export SESSD_BENCH_VERBOSE=1
timeout -k 5 300 python -u bench.py --steps 200 --warmup 20 ${BENCH_ARGS:-} 2> gpurun_out/r3_${TAG}_bench.err | tail -1 > gpurun_out/r3_${TAG}_bench.json; echo "bench exit $?"
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3_${TAG}_bench.json'))
    print('value', round(d['value'],1), 'seq', d.get('value_sequential',{}).get('frames_per_s'), 'stages', d.get('stages_ms_eager'))
    p=d.get('parity') or {}
    print('parity', {k:p.get(k) for k in ('frames','identical','flipped_near_threshold','ok','bev_rel_err')}, [m['why'][:120] for m in p.get('mismatch',[])][:3])
    r=d.get('roofline',{}); print('roofline', r.get('avg_launch_ms'), r.get('frac'), r.get('dense_launch_ms'))
    print('host_io', {k:v for k,v in d.get('host_io',{}).items() if 'frames_per_s' in k})
    m=d.get('roofline_spmiddle',{}); print('spmiddle', m.get('ms'), m.get('frac'), m.get('mfma',{}).get('conv_ms'), [round(l['ms']*1e3,1) for l in m.get('mfma',{}).get('layers',[])])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r3_${TAG}_bench.err').read()[-2000:])
PY
fi
if [ -n "${TRACE:-}" ]; then
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o t -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 --no-roofline --no-host-io --no-sequential ${BENCH_ARGS:-} > $R/gpurun_out/prof_$TAG.log 2>&1
DB=$(find $R/gpurun_out/prof_$TAG -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 100 50 > $R/gpurun_out/r3_${TAG}_trace_1stream.txt; head -48 $R/gpurun_out/r3_${TAG}_trace_1stream.txt | cut -c1-60,100-160
rm -rf $R/gpurun_out/prof_$TAG
fi
