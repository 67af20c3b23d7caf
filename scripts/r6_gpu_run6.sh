#!/bin/bash
# round 6: counter passes of the dense stage with a FIXED configuration (no autotune: every pass profiles the same launches), on a
# CU-masked half and on the whole chip; the random-weights line under the per-frame code-scale rule; the frame-10 probe
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
for mode in half whole; do
  [ $mode = half ] && H="--cu-half" || H=""
  files=""; i=0
  for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D=$O/dense_pmc_${mode}_$i
    rm -rf $D
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --fixed $H > $O/dense_pmc_${mode}_$i.log 2>&1
    echo "dense pmc ($mode) pass $i rc $?"
    f=$(find $D -name "*counter_collection.csv" | head -1)
    files="$files $f"
    [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
  done
  python $R/scripts/pmc_compact.py "WHOLE FRAME, batch 1, the engine on $mode chip, FIXED configuration (engine.force_active_tiles(), whole-unit list shares: scripts/sparse_probe.py --fixed), three separate --pmc passes over the same launches" $files --trace $tr --tail 400 > $O/frame_pmc_$mode.txt
  grep "fixed configuration\|stages\|sites" $O/dense_pmc_${mode}_1.log | sed 's/^/# /' >> $O/frame_pmc_$mode.txt
  for i in 1 2 3; do rm -rf $O/dense_pmc_${mode}_$i; done
  cut -c1-170 $O/frame_pmc_$mode.txt | head -50
done
cd $R
python scripts/r6_traffic_json.py $O/frame_pmc_half.txt $O/frame_pmc_whole.txt > $O/r6_wino_traffic.json; grep -n "times_algorithmic\|traffic_configuration\|write_at_least" $O/r6_wino_traffic.json
timeout -k 5 600 python bench.py --random-weights --no-train-step --no-host-io --no-sequential > $O/bench_random.json 2>$O/bench_random.err; echo "random rc $?"
timeout -k 5 600 python scripts/parity_case_probe.py > $O/parity_case_probe.json 2>$O/parity_case_probe.err; echo "probe rc $?"; head -c 1500 $O/parity_case_probe.json
