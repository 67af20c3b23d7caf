#!/bin/bash
# round 6, end-of-round artefacts on ONE box: the whole GPU suite, smoke(), the bench lines (driver command = in-process trained
# weights + strict gate; 300-step default; one stream; two plain streams = rounds 1 - 4; stress; random weights), kernel traces of the timed
# regions and of 40 replays of the training iteration
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6z; mkdir -p $O $R/build
cd $R
timeout -k 5 1800 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-300
W=build/r6_student.pt
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --save-weights $W > $O/bench_driver_command.json 2>$O/bench_driver_command.err; echo "driver-line rc $?"
timeout -k 5 600 python bench.py > $O/bench_default_trains_itself.json 2>$O/bench_default_trains_itself.err; echo "default (trains itself) rc $?"
timeout -k 5 600 python bench.py --weights $W --no-train-step > $O/bench_4_in_flight.json 2>$O/bench_4_in_flight.err; echo "4 in flight rc $?"
timeout -k 5 600 python bench.py --weights $W --streams 1 --no-train-step > $O/bench_1stream.json 2>$O/bench_1stream.err; echo "1stream rc $?"
timeout -k 5 600 python bench.py --weights $W --streams 2 --cu-split none --no-train-step > $O/bench_two_plain_streams.json 2>$O/bench_two_plain_streams.err; echo "two plain streams rc $?"
timeout -k 5 600 python bench.py --stress --no-train-step > $O/bench_stress_batch8.json 2>$O/bench_stress_batch8.err; echo "stress rc $?"
timeout -k 5 600 python bench.py --random-weights --no-train-step > $O/bench_random_weights.json 2>$O/bench_random_weights.err; echo "random rc $?"
timeout -k 5 600 python bench.py --weights $W --no-train-step --no-active-conv > $O/bench_4_in_flight_full_map_convs.json 2>$O/bench_4_in_flight_full_map_convs.err; echo "4 in flight, conv_0 / conv_1 as the full-map launch rc $?"
python - <<'PY'
import json
for n in ("driver_command", "default_trains_itself", "4_in_flight", "4_in_flight_full_map_convs", "1stream", "two_plain_streams", "stress_batch8", "random_weights"):
    try:
        d = json.loads(open("gpurun_out/r6z/bench_%s.json" % n).read().strip().splitlines()[-1])
        c, r = d["config"], d.get("roofline") or {}
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), c["frames_in_flight"], c.get("parity_ok"), c.get("parity_matched"), c.get("parity_frames"), c.get("parity_rule"), c.get("weights"),
              "frac", round(r.get("frac", 0), 4), round(r.get("frac_of_cu_set_peak", 0), 4), round(r.get("frac_chip_timed_region", 0), 4), r.get("frac_chip_timed_region_from_counters"), round(r.get("frac_full_map_launches", 0), 4), r.get("frac_list_launches"),
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
# counter passes of whole frames of the FIXED configuration (the same launches in every pass), CU-masked half and whole chip
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
for mode in half whole; do
  [ $mode = half ] && H="--cu-half" || H=""
  files=""; i=0
  for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D=$O/frame_pmc_${mode}_$i
    rm -rf $D
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --fixed $H > $O/frame_pmc_${mode}_$i.log 2>&1
    echo "frame pmc ($mode) pass $i rc $?"
    f=$(find $D -name "*counter_collection.csv" | head -1)
    files="$files $f"
    [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
  done
  python $R/scripts/pmc_compact.py "WHOLE FRAME, batch 1, the engine on $mode chip, FIXED configuration (engine.force_active_tiles(), whole-unit list shares: scripts/sparse_probe.py --fixed), three separate --pmc passes over the same launches" $files --trace $tr --tail 400 > $O/frame_pmc_$mode.txt
  grep "fixed configuration\|stages\|sites\|active_tile_fractions" $O/frame_pmc_${mode}_1.log | sed 's/^/# /' >> $O/frame_pmc_$mode.txt
  for i in 1 2 3; do rm -rf $O/frame_pmc_${mode}_$i; done
done
python $R/scripts/r6_traffic_json.py $O/frame_pmc_half.txt $O/frame_pmc_whole.txt > $O/r6_wino_traffic.json; grep -n "times_algorithmic\|traffic_configuration" $O/r6_wino_traffic.json
python $R/scripts/r6_mfma_flops_json.py $O/frame_pmc_half.txt $O/frame_pmc_whole.txt > $O/r6_mfma_flops_per_frame.json; grep -n "executed_gflop_per_frame" $O/r6_mfma_flops_per_frame.json
rm -rf $O/p_train
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_train -o tr -- python $R/scripts/train_step_bench.py --real-loss --replays-only 40 > $O/p_train.log 2>&1
echo "train replay rc $?"; tail -1 $O/p_train.log | cut -c1-300
DB=$(find $O/p_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 40 140 sparse_pack_batch_kernel > $O/trace_train_replay.txt; head -3 $O/trace_train_replay.txt | cut -c1-150; tail -1 $O/trace_train_replay.txt
rm -rf $O/p_train
