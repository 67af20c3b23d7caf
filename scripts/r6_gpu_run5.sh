#!/bin/bash
# round 6: validation of the state after the measurement changes -- whole GPU suite, smoke(), the bench lines (driver command on
# in-process trained weights, default, one stream, stress, random weights), kernel traces of the timed regions, counter passes of the
# dense stage on a CU-masked half AND on the whole chip with the launches of KNOWN bytes (fill kernels) for the WRITE_SIZE calibration,
# the trained-parity record of seed 1 again (round 5's carried sparse_overflow_flag = 1)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e; mkdir -p $O $R/build
cd $R
timeout -k 5 1800 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-300
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --save-weights build/r6_student.pt > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver-line rc $?"
W=build/r6_student.pt
timeout -k 5 600 python bench.py --weights $W --no-train-step > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"
timeout -k 5 600 python bench.py --weights $W --streams 1 --no-train-step > $O/bench_1stream.json 2>$O/bench_1stream.err; echo "1stream rc $?"
timeout -k 5 600 python bench.py --stress --no-train-step > $O/bench_stress.json 2>$O/bench_stress.err; echo "stress rc $?"
timeout -k 5 600 python bench.py --random-weights --no-train-step > $O/bench_random.json 2>$O/bench_random.err; echo "random rc $?"
timeout -k 5 600 python bench.py --weights $W --streams 2 --cu-split none --no-train-step > $O/bench_r4config.json 2>$O/bench_r4config.err; echo "r4 config rc $?"
python - <<'PY'
import json
for n in ("driver", "default", "1stream", "stress", "random", "r4config"):
    try:
        d = json.loads(open("gpurun_out/r6e/bench_%s.json" % n).read().strip().splitlines()[-1])
        c, r = d["config"], d.get("roofline") or {}
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), c["frames_in_flight"], c.get("parity_ok"), c.get("parity_matched"), c.get("parity_frames"), c.get("parity_rule"), c.get("weights"),
              "frac", r.get("frac"), r.get("frac_of_cu_set_peak"), r.get("frac_chip_timed_region"), r.get("frac_full_map_launches"), r.get("frac_list_launches"),
              (d.get("roofline_whole_chip_engine") or {}).get("frac"), d.get("stages_ms_eager"), c.get("value_sequential_frames_per_s"),
              {k: (d.get("train_step") or {}).get(k) for k in ("ms_per_iter", "ms_per_iter_fresh_batches", "matched_boxes", "sparse_overflow_flag")},
              {k: (d.get("host_io") or {}).get(k) for k in ("frames_per_s", "latency_mode_frames_per_s")}, round((d.get("roofline_spmiddle") or {}).get("frac", 0), 4), c.get("seconds_to_first_timed_step"))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
cd /tmp && export TMPDIR=/tmp
for cfg in 4inflight 1stream stress; do
  case $cfg in
    1stream)  A="--weights $R/$W --steps 100 --warmup 10 --streams 1"; F=100;;
    4inflight) A="--weights $R/$W --steps 400 --warmup 40"; F=400;;
    stress)   A="--stress --steps 30 --warmup 5"; F=30;;
  esac
  rm -rf $O/p_$cfg
  timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_$cfg -o t -- python $R/bench.py $A --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p_$cfg.log 2>&1
  echo "$cfg trace rc $?"
  DB=$(find $O/p_$cfg -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $F 60 > $O/trace_$cfg.txt; head -3 $O/trace_$cfg.txt | cut -c1-150
  rm -rf $O/p_$cfg
done
# counters of the dense stage: CU-masked half (the timed configuration) and whole chip, fill launches (known bytes) included
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
for mode in half whole; do
  [ $mode = half ] && H="--cu-half" || H=""
  files=""; i=0
  for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D=$O/dense_pmc_${mode}_$i
    rm -rf $D
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --force-active --list-shares whole $H > $O/dense_pmc_${mode}_$i.log 2>&1
    echo "dense pmc ($mode) pass $i rc $?"
    f=$(find $D -name "*counter_collection.csv" | head -1)
    files="$files $f"
    [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
  done
  python $R/scripts/pmc_compact.py "SSFA neck + heads + the frame's fill launches, batch 1, the engine on $mode chip, active-tile mode, whole-unit list shares" $files --trace $tr --tail 400 --match winograd --match conv2d_sk --match conv2d_mfma --match bev_tile --match fill_ --match ssfa_fuse > $O/dense_pmc_$mode.txt
  grep "active_tiles\|stages" $O/dense_pmc_${mode}_1.log | sed 's/^/# /' >> $O/dense_pmc_$mode.txt
  for i in 1 2 3; do rm -rf $O/dense_pmc_${mode}_$i; done
  cut -c1-170 $O/dense_pmc_$mode.txt | head -20
done
cd $R
timeout -k 5 900 python tests/trained_parity.py --iterations 2000 --scenes 400 --heldout 200 --seed 1 --out $O/trained_parity_seed1.json > $O/trained_parity_seed1.log 2>&1; echo "trained parity seed 1 rc $?"; tail -25 $O/trained_parity_seed1.log | cut -c1-200
