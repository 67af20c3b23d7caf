#!/bin/bash
# round 5, GPU call 2: boundary-vs-barrier ubench; tests of the changed kernels (whole-unit shares, four-wave tile activity);
# A/B of the list-launch share modes on ONE box (two frames in flight and sequential); counter passes of the list launches
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/bvb scripts/ubench/boundary_vs_barrier.hip > $O/ubench_build.log 2>&1
timeout -k 5 120 /tmp/bvb > $O/boundary_vs_barrier.txt 2>&1; echo "ubench rc $?"; cat $O/boundary_vs_barrier.txt
timeout -k 5 900 python -m pytest tests/test_dense_active_gpu.py tests/test_active_rule_cpu.py -x -q -m gpu > $O/tests_active.log 2>&1; echo "tests_active rc $?"; tail -3 $O/tests_active.log
timeout -k 5 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "active or autotune or stress" > $O/tests_pipe.log 2>&1; echo "tests_pipe rc $?"; tail -2 $O/tests_pipe.log
for mode in cut auto whole cut whole; do
  for st in 2 1; do
    n=$(ls $O/ab_${mode}_s${st}_*.json 2>/dev/null | wc -l)
    timeout -k 5 300 python bench.py --steps 400 --warmup 40 --streams $st --list-shares $mode --cpu-frames 8 --no-host-io --no-sequential --no-train-step > $O/ab_${mode}_s${st}_$n.json 2>$O/ab_${mode}_s${st}_$n.err
    echo "ab $mode streams $st rc $?"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5b/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], "frac", round(r["frac"], 3), "list", r["frac_list_launches"] and round(r["frac_list_launches"], 3),
              {k: round(v * 1e3, 1) for k, v in r["dense_launch_ms"].items()}, d["stages_ms_eager"],
              {k: (v.get("min_rounds"), v.get("streamk_shape")) for k, v in d["config"]["tuning"]["active_tiles"].items()})
    except Exception as ex:
        print(f, "unreadable", ex)
PY
# counters of the dense stage with the list layers on whole-unit shares and on stream-K shares
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
for mode in whole cut; do
  files=""; i=0
  for set in "$SQ1" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D=$O/dense_pmc_${mode}_$i
    rm -rf $D
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python $R/scripts/sparse_probe.py --frames 3 --force-active --list-shares $mode > $O/dense_pmc_${mode}_$i.log 2>&1
    echo "dense pmc $mode pass $i rc $?"
    f=$(find $D -name "*counter_collection.csv" | head -1)
    files="$files $f"
    [ $i = 1 ] && tr=$(find $D -name "*kernel_trace.csv" | head -1)
  done
  python $R/scripts/pmc_compact.py "SSFA neck + heads, batch 1, active-tile mode, Winograd list layers on $mode shares" $files --trace $tr --tail 400 --match winograd --match conv2d_sk --match conv2d_mfma --match bev_tile --match fill_inactive --match ssfa_fuse > $O/dense_pmc_${mode}.txt
  grep "active_tiles\|stages" $O/dense_pmc_${mode}_1.log | sed 's/^/# /' >> $O/dense_pmc_${mode}.txt
  for i in 1 2 3; do rm -rf $O/dense_pmc_${mode}_$i; done
  cut -c1-170 $O/dense_pmc_${mode}.txt | head -24
done
