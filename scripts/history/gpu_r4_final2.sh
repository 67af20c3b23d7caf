#!/bin/bash
# round 4, second session: full validation of the final state -- the whole GPU suite, smoke(), the bench lines and kernel traces
# that go into profiles/, counter passes of the dense stage with the active-tile kernels
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f2; mkdir -p $O
cd $R
timeout -k 5 1200 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; echo "driver-line rc $?"
timeout -k 5 600 python bench.py --no-train-step > $O/bench_default.json 2>$O/bench_default.err; echo "default rc $?"
timeout -k 5 600 python bench.py --streams 1 --no-train-step > $O/bench_1stream.json 2>$O/bench_1stream.err; echo "1stream rc $?"
timeout -k 5 600 python bench.py --stress --no-train-step > $O/bench_stress.json 2>$O/bench_stress.err; echo "stress rc $?"
python - <<'PY'
import json
for n in ("driver","default","1stream","stress"):
    try:
        d=json.loads(open("gpurun_out/r4f2/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d["parity"].get("ok"), d["parity"].get("identical"), round(d["roofline"]["frac"],3), d["roofline"].get("frac_full_map_launches"), d.get("stages_ms_eager"), (d.get("value_sequential") or {}).get("frames_per_s"), (d.get("train_step") or {}).get("ms_per_iter"), (d.get("host_io") or {}))
    except Exception as ex:
        print(n, "unreadable", ex)
PY
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
# counters of the dense stage as the frame runs it (eager frames after the autotune sweep)
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
files=""; i=0
for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$O/dense_pmc$i
  rm -rf $D
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --force-active > $O/dense_pmc$i.log 2>&1
  echo "dense pmc pass $i rc $?"
  f=$(find $D -name "*counter_collection.csv" | head -1)
  files="$files $f"
  [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
done
python $R/scripts/pmc_compact.py "SSFA neck + heads, batch 1, blocks 0 / 1, trans_0 and the transposed pair in active-tile mode" $files --trace $tr --tail 400 --match winograd --match conv2d_sk --match conv2d_mfma --match bev_tile --match fill_inactive --match ssfa_fuse > $O/dense_pmc_summary.txt
for i in 1 2 3; do rm -rf $O/dense_pmc$i; done
cut -c1-170 $O/dense_pmc_summary.txt | head -30
# the training iteration (driver key train_step) once, with its trace
cd $R
timeout -k 5 400 python bench.py --steps 20 --warmup 5 --cpu-frames 8 --no-host-io --no-sequential --no-roofline > $O/bench_train.json 2>$O/bench_train.err; echo "train-step line rc $?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4f2/bench_train.json").read().strip().splitlines()[-1])
    print("train_step", d.get("train_step"))
except Exception as ex:
    print("unreadable", ex)
PY
