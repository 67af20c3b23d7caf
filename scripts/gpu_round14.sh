R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 400 python scripts/train_step_bench.py --cpu 2>/dev/null | tail -1 | tee gpurun_out/train_step.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o tr -- python $R/scripts/train_step_bench.py --steps 5 > $R/gpurun_out/prof_train.log 2>&1
DB=$(find $R/gpurun_out/prof_train -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB 0 30 > $R/gpurun_out/prof_train_summary.txt; head -34 $R/gpurun_out/prof_train_summary.txt | cut -c1-100,110-170
