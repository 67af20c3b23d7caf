#!/bin/bash
# round 4, end of round: kernel traces of the timed configurations (1 stream, 2 streams, dense-scene batch, training replay)
# + the float32 F(4x4,3x3) accuracy probe (CPU arithmetic; the GPU only calibrates the seeded model)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4tr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in 1stream 2streams stress; do
  case $cfg in
    1stream)  A="--steps 100 --warmup 10 --streams 1"; F=100;;
    2streams) A="--steps 200 --warmup 20 --streams 2"; F=200;;
    stress)   A="--stress --steps 30 --warmup 5"; F=30;;
  esac
  rm -rf $O/p_$cfg
  timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_$cfg -o t -- python $R/bench.py $A --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p_$cfg.log 2>&1
  echo "$cfg rc $?"
  DB=$(find $O/p_$cfg -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $F 60 > $O/trace_$cfg.txt; head -3 $O/trace_$cfg.txt | cut -c1-150
  rm -rf $O/p_$cfg
done
rm -rf $O/p_train
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/p_train -o tr -- python $R/scripts/train_step_bench.py --real-loss --replays-only 40 > $O/p_train.log 2>&1
echo "train rc $?"
DB=$(find $O/p_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 40 70 sparse_pack_batch_kernel > $O/trace_train_replay.txt; head -3 $O/trace_train_replay.txt | cut -c1-150
rm -rf $O/p_train
cd $R
timeout -k 5 600 python scripts/wino_f43_accuracy_probe.py > $O/f43.json 2>$O/f43.err; echo "f43 rc $?"; head -c 3000 $O/f43.json
