#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6l; mkdir -p $O $R/build; cd $R
W=build/r6_student.pt
[ -f $W ] || timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-host-io --no-sequential --no-roofline --cpu-frames 4 --save-weights $W > $O/train.json 2>$O/train.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python $R/bench.py --weights $R/$W --steps 600 --warmup 40 --cpu-frames 0 --no-roofline --no-host-io --no-sequential --no-train-step > $O/p.log 2>&1; echo "trace rc $?"
CSV=$(find $O/p -name "*kernel_trace.csv" | head -1)
python $R/scripts/r6_set_timeline.py $CSV 400 > $O/set_timeline.json; cat $O/set_timeline.json
rm -rf $O/p
