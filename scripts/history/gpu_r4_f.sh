#!/bin/bash
# round 4, sixth GPU call: sorted-tile probe, kernel traces (training replay with the real loss, inference one stream)
R=$GRAFT_REPO_ROOT
cd $R || exit 1
mkdir -p gpurun_out/r4f
export TMPDIR=/tmp
timeout 400 python scripts/sorted_tiles_probe.py > gpurun_out/r4f/sorted_tiles_probe.json 2> gpurun_out/r4f/sorted_tiles_probe.err
echo "probe rc $?"; python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4f/sorted_tiles_probe.json"))
    for k, v in j.items():
        print(k, "plain", v["plain_conv_ms"], "sorted", v["sorted_conv_ms"], "useful", v["plain_useful"], v["sorted_useful"])
        print("   ", [(L["layer"], L["plain_ms"], L["sorted_ms"], L["plain_steps"], L["sorted_steps"]) for L in v["layers"]])
except Exception as e:
    print("unreadable", e)
PY
cd /tmp
rm -rf $R/gpurun_out/prof_replay
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_replay -o tr -- python $R/scripts/train_step_bench.py --real-loss --replays-only 40 > $R/gpurun_out/r4f/prof_replay.log 2>&1
tail -1 $R/gpurun_out/r4f/prof_replay.log | cut -c1-300
DB=$(find $R/gpurun_out/prof_replay -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 40 150 sparse_pack_batch_kernel fill_u32_kernel > $R/gpurun_out/r4f/train_replay_trace.txt; head -3 $R/gpurun_out/r4f/train_replay_trace.txt | cut -c1-150; tail -12 $R/gpurun_out/r4f/train_replay_trace.txt | cut -c1-160
rm -rf $R/gpurun_out/prof_replay
rm -rf $R/gpurun_out/prof_f1
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f1 -o f1 -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --streams 1 --no-roofline --no-host-io --no-sequential --no-train-step > $R/gpurun_out/r4f/prof_f1.log 2>&1
DB=$(find $R/gpurun_out/prof_f1 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 100 60 > $R/gpurun_out/r4f/trace_1stream.txt; head -3 $R/gpurun_out/r4f/trace_1stream.txt | cut -c1-150
rm -rf $R/gpurun_out/prof_f1
