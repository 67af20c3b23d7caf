#!/bin/bash
# round 6 (review item 6): the reference's TEST batch (tools/test.py:211: samples_per_gpu = 4) and batch x CU sets, informational --
# KITTI-size frames (20 k points, max 16000 voxels per frame), trained weights, strict gate. The headline stays configs[1] (batch-1 engines).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6sweep; mkdir -p $O
cd $R
W=${WEIGHTS:-build/r6_student.pt}
[ -f $W ] || timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-host-io --no-sequential --no-roofline --cpu-frames 4 --save-weights $W > $O/train.json 2>$O/train.err
B="--weights $W --no-train-step --no-host-io --cpu-frames 16 --steps 200 --warmup 20"
run() { n=$1; shift; timeout -k 5 500 python bench.py $B "$@" > $O/$n.json 2>$O/$n.err; echo "$n rc $?"; }
run s4_b1_halves --streams 4 --batch 1 --steps 400 --warmup 40
run s1_b4_whole  --streams 1 --batch 4
run s2_b2_halves --streams 2 --batch 2
run s2_b4_halves --streams 2 --batch 4
run s4_b2_halves --streams 4 --batch 2
run s1_b2_whole  --streams 1 --batch 2
run s2_b2_plain  --streams 2 --batch 2 --cu-split none
python - <<'PY'
import json, glob, os
rows = []
for f in sorted(glob.glob("gpurun_out/r6sweep/s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        rows.append({"run": os.path.basename(f)[:-5], "error": repr(ex), "stderr_tail": open(f[:-5] + ".err").read()[-300:]})
        continue
    c, r, sp = d["config"], d.get("roofline") or {}, d.get("roofline_spmiddle") or {}
    B = int(c["workload"].split("batch ")[1].split(" ")[0])
    rows.append({"run": os.path.basename(f)[:-5], "engines": c["frames_in_flight"], "batch": B, "cu_sets": c["cu_sets"], "cus_per_set": c["cus_per_set"],
                 "frames_per_s": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 4),
                 "ms_latency_per_batch": round(c["ms_latency_per_frame_in_flight"] * B, 3),
                 "parity_ok": c.get("parity_ok"), "parity_matched": c.get("parity_matched"), "parity_frames": c.get("parity_frames"), "parity_rule": c.get("parity_rule"),
                 "roofline_frac": r.get("frac"), "frac_of_cu_set_peak": r.get("frac_of_cu_set_peak"), "frac_full_map_launches": r.get("frac_full_map_launches"),
                 "frac_list_launches": r.get("frac_list_launches"), "frac_list_launches_of_cu_set_peak": r.get("frac_list_launches_of_cu_set_peak"),
                 "roofline_spmiddle_frac": sp.get("frac"), "spmiddle_ms": sp.get("ms"),
                 "spmiddle_executed_frac_of_f32_mfma_peak": (sp.get("mfma") or {}).get("executed_frac_of_f32_mfma_peak"),
                 "spmiddle_useful_row_fraction": (sp.get("mfma") or {}).get("useful_row_fraction"),
                 "stages_ms_eager": d.get("stages_ms_eager")})
    print(rows[-1])
json.dump({"what": "round 6, review item 6: KITTI-size frames (20 k points) at the reference's test batch (tools/test.py:211: 4) and batch x CU sets; "
                   "trained weights, strict gate; every row one `python bench.py --weights ... --streams S --batch B` on ONE box (scripts/r6_batch_cu_sweep.sh). "
                   "frames_per_s = engines' whole-job rate; roofline / stage figures are bench.py's legs on engine 0 of the configuration. Informational: the headline "
                   "stays BASELINE configs[1], batch-1 engines.", "rows": rows}, open("gpurun_out/r6sweep/r6_batch_cu_sweep.json", "w"), indent=1)
PY
