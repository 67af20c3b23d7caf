#!/bin/bash
# round 5 EXPERIMENT: frames in flight on CU-masked streams (each frame its own compute units) against shared streams
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
run() { # name, args
  timeout -k 5 300 python bench.py --steps 400 --warmup 40 --cpu-frames 8 --no-host-io --no-sequential --no-train-step $2 > $O/$1.json 2>$O/$1.err; echo "$1 rc $?"
}
run base_s2 "--streams 2"
run split2_contig "--streams 2 --cu-split contiguous"
run split2_inter "--streams 2 --cu-split interleaved"
run split4_contig "--streams 4 --cu-split contiguous"
run split4_inter "--streams 4 --cu-split interleaved"
run split8_contig "--streams 8 --cu-split contiguous"
run base_s4 "--streams 4"
run split2_contig_eager "--streams 2 --cu-split contiguous --eager"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5h/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("/")[-1], round(d["value"], 1), d["parity"]["ok"], d["parity"]["identical"], d["config"].get("cus_per_frame_in_flight"),
              {k: round(v * 1e3, 1) for k, v in (r.get("dense_launch_ms") or {}).items()}, d.get("stages_ms_eager"))
    except Exception as ex:
        print(f, "unreadable", ex, open(f.replace(".json", ".err")).read()[-400:])
PY
