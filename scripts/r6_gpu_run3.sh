#!/bin/bash
# round 6, first call: the whole GPU suite with the new tests (RCCL on masked streams, ordered teardown, corner range filter, sticky
# sparse-overflow flags), the sparse-overflow scan of the training batches, the driver's command on in-process TRAINED weights (strict
# gate) and the round-5 line beside it, a kernel trace of the timed region, and the first throughput-regime A/Bs of the sparse levers
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c; mkdir -p $O $R/build
cd $R
export SESSD_BENCH_VERBOSE=1
timeout -k 5 1500 python -m pytest tests/test_trained_gpu.py tests/test_trainloop_gpu.py tests/test_train_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log
timeout -k 5 400 python scripts/r6_sparse_overflow_scan.py 64 300 > $O/overflow_scan.log 2>&1; echo "scan rc $?"; grep -v Warn $O/overflow_scan.log | tail -3 | cut -c1-400
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --save-weights build/r6_student.pt > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver rc $?"; tail -3 $O/bench_driver.err | cut -c1-400
timeout -k 5 600 python bench.py --random-weights --no-train-step --no-host-io > $O/bench_random.json 2>$O/bench_random.err; echo "random rc $?"
B="--weights build/r6_student.pt --no-train-step --no-host-io --no-sequential --no-roofline --cpu-frames 8 --steps 400 --warmup 40"
run() { n=$1; shift; timeout -k 5 400 env "$@" python bench.py $B $EXTRA > $O/ab_$n.json 2>$O/ab_$n.err; echo "$n rc $?"; }
EXTRA="" run base_a X=1
EXTRA="--sort-tiles" run sort_auto X=1
EXTRA="--sort-tiles" run sort_l2 SESSD_FORCE_SPARSE="6::1,7::1,8::1"
EXTRA="--sort-tiles" run sort_all SESSD_FORCE_SPARSE="0::1,1::1,2::1,3::1,4::1,5::1,6::1,7::1,8::1,9::1,10::1,11::1,12::1"
EXTRA="--sparse-mt" run mt_auto X=1
EXTRA="--no-offset-split" run noksplit X=1
EXTRA="" run base_b X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]
        print(f.split("/")[-1], round(d["value"], 1), c.get("parity_ok"), c.get("parity_matched"), c.get("parity_frames"), c.get("parity_rule"), c.get("weights"),
              (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("frac_of_cu_set_peak"), {k: v for k, v in (d.get("train_step") or {}).items() if k in ("ms_per_iter", "ms_per_iter_fresh_batches", "sparse_overflow_flag", "error")},
              d["tuning"]["sparse"], d["tuning"]["sparse_offset_pattern_tiles"])
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-400:])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_4
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_4 -o t -- python $R/bench.py --steps 400 --warmup 40 --weights $R/build/r6_student.pt --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p_4.log 2>&1
echo "trace rc $?"
DB=$(find $O/p_4 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 400 60 > $O/trace_4inflight.txt; head -50 $O/trace_4inflight.txt | cut -c1-180
rm -rf $O/p_4
