#!/bin/bash
# round 5: full validation of the final state -- the whole GPU suite, smoke(), the bench lines and kernel traces that go into
# profiles/, the trained-weights line (strict gate), the sparse stage's counters on the dense-scene batch
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
timeout -k 5 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver-line rc $?"
timeout -k 5 600 python bench.py --no-train-step > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"
timeout -k 5 600 python bench.py --streams 1 --no-train-step > $O/bench_1stream.json 2>$O/bench_1stream.err; echo "1stream rc $?"
timeout -k 5 600 python bench.py --stress --no-train-step > $O/bench_stress.json 2>$O/bench_stress.err; echo "stress rc $?"
timeout -k 5 600 python bench.py --streams 2 --cu-split none --no-train-step > $O/bench_r4config.json 2>$O/bench_r4config.err; echo "r4 config (two plain streams) rc $?"
if [ -f build/r5_trained_student.pt ]; then
  timeout -k 5 600 python bench.py --weights build/r5_trained_student.pt --no-train-step > $O/bench_trained.json 2>$O/bench_trained.err; echo "trained rc $?"
fi
python - <<'PY'
import json
for n in ("driver", "default", "1stream", "stress", "r4config", "trained"):
    try:
        d = json.loads(open("gpurun_out/r5f/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), d["config"]["frames_in_flight"], d["parity"].get("ok"), d["parity"].get("identical"), d["parity"].get("frames"), d["parity"].get("rule_set"), round(d["roofline"]["frac"], 3), d["roofline"].get("frac_of_whole_chip_peak"), (d.get("roofline_whole_chip_engine") or {}).get("frac"),
              d["roofline"].get("frac_full_map_launches"), d["roofline"].get("frac_list_launches"), d.get("stages_ms_eager"), (d.get("value_sequential") or {}).get("frames_per_s"),
              {k: (d.get("train_step") or {}).get(k) for k in ("ms_per_iter", "ms_per_iter_fresh_batches", "matched_boxes")}, {k: (d.get("host_io") or {}).get(k) for k in ("frames_per_s", "latency_mode_frames_per_s")},
              round(d["roofline_spmiddle"]["frac"], 4))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
cd /tmp && export TMPDIR=/tmp
for cfg in 1stream 4inflight 2streams stress; do
  case $cfg in
    1stream)  A="--steps 100 --warmup 10 --streams 1"; F=100;;
    4inflight) A="--steps 400 --warmup 40"; F=400;;
    2streams) A="--steps 200 --warmup 20 --streams 2 --cu-split none"; F=200;;
    stress)   A="--stress --steps 30 --warmup 5"; F=30;;
  esac
  rm -rf $O/p_$cfg
  timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_$cfg -o t -- python $R/bench.py $A --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p_$cfg.log 2>&1
  echo "$cfg rc $?"
  DB=$(find $O/p_$cfg -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $F 60 > $O/trace_$cfg.txt; head -3 $O/trace_$cfg.txt | cut -c1-150
  rm -rf $O/p_$cfg
done
# the captured training iteration alone (40 replays)
rm -rf $O/p_train
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_train -o tr -- python $R/scripts/train_step_bench.py --real-loss --replays-only 40 > $O/p_train.log 2>&1
echo "train replay rc $?"; tail -1 $O/p_train.log | cut -c1-300
DB=$(find $O/p_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 40 140 sparse_pack_batch_kernel > $O/trace_train_replay.txt; head -3 $O/trace_train_replay.txt | cut -c1-150; tail -1 $O/trace_train_replay.txt
rm -rf $O/p_train
# counters of the sparse stage, dense-scene batch (after the bucketed hash: chain_rulebook / vox_insert)
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
files=""; i=0
for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$O/stress_pmc$i
  rm -rf $D
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --stress --frames 3 > $O/stress_pmc$i.log 2>&1
  echo "stress pmc pass $i rc $?"
  f=$(find $D -name "*counter_collection.csv" | head -1)
  files="$files $f"
  [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
done
python $R/scripts/pmc_compact.py "SpMiddleFHD + voxelizer, dense-scene batch (8 x 200 k points)" $files --trace $tr --tail 420 --match sparse_conv --match chain_ --match vox_ > $O/sparse_pmc_stress.txt
grep "sites\|stages" $O/stress_pmc1.log | sed 's/^/# /' >> $O/sparse_pmc_stress.txt
for i in 1 2 3; do rm -rf $O/stress_pmc$i; done
cut -c1-170 $O/sparse_pmc_stress.txt | head -30
# counters of the dense stage in the TIMED configuration: the engine on a CU-masked half, whole-unit list shares
cd /tmp && export TMPDIR=/tmp
files=""; i=0
for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$O/dense_pmc_half_$i
  rm -rf $D
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --force-active --list-shares whole --cu-half > $O/dense_pmc_half_$i.log 2>&1
  echo "dense pmc (CU half) pass $i rc $?"
  f=$(find $D -name "*counter_collection.csv" | head -1)
  files="$files $f"
  [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
done
python $R/scripts/pmc_compact.py "SSFA neck + heads, batch 1, the engine on ONE HALF of the chip (CU-masked stream, 128-CU launches), active-tile mode, whole-unit list shares" $files --trace $tr --tail 400 --match winograd --match conv2d_sk --match conv2d_mfma --match bev_tile --match fill_inactive --match ssfa_fuse > $O/dense_pmc_cu_half.txt
grep "active_tiles\|stages" $O/dense_pmc_half_1.log | sed 's/^/# /' >> $O/dense_pmc_cu_half.txt
for i in 1 2 3; do rm -rf $O/dense_pmc_half_$i; done
cut -c1-170 $O/dense_pmc_cu_half.txt | head -20
