"""Round 6: what each CU set does over time in the timed configuration, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps N` (four engines, engine i on set i % 2; a hardware queue per engine):
    python scripts/r6_set_timeline.py kernel_trace.csv [last_frames]
Per set: share of the wall time with no kernel running, one kernel (by class), two kernels (by class pair); and the mean duration of
the dense launches when they run alone on the set against when the other engine's kernel runs beside them."""
import csv
import json
import sys
from collections import defaultdict

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 400
starts = [r[0] for r in rows if "vox_insert_kernel" in r[2]]
t_lo, t_hi = starts[-nlast - 8], starts[-8]
rows = [r for r in rows if t_lo <= r[0] < t_hi]
queues = sorted({r[3] for r in rows if "vox_insert_kernel" in r[2]}, key=lambda q: min(r[0] for r in rows if r[3] == q and "vox_insert_kernel" in r[2]))
# engine order = the order the queues were created in = ascending queue id in practice; engine i runs on set i % 2
queues = sorted(queues, key=lambda q: int(q))
set_of = {q: i % 2 for i, q in enumerate(queues)}


def cls(n):
    if "winograd" in n or "conv2d_sk" in n or "conv2d_mfma" in n:
        return "dense"
    if "ssfa_fuse" in n or "fill_inactive" in n or "fill_multi" in n:
        return "hbm"
    if "sparse_conv" in n:
        return "sparse"
    return "small"


out = {"queues": queues, "frames": nlast, "wall_us_per_frame": (t_hi - t_lo) / 1e3 / nlast}
for s in (0, 1):
    ev = []
    for a, b, n, q in rows:
        if set_of.get(q) == s:
            ev.append((a, 1, cls(n), q))
            ev.append((b, -1, cls(n), q))
    ev.sort()
    live, acc, last = defaultdict(int), defaultdict(float), ev[0][0]
    for t, d, c, q in ev:
        if t > last:
            k = sorted(x for x, v in live.items() for _ in range(v))
            acc["idle" if not k else "+".join(k)] += t - last
        live[c] += d
        last = t
    tot = sum(acc.values())
    out["set%d" % s] = {k: round(100.0 * v / tot, 1) for k, v in sorted(acc.items(), key=lambda x: -x[1])}
    out["set%d_wall_us_per_frame_of_the_set" % s] = round(tot / 1e3 / (nlast / 2.0), 1)
print(json.dumps(out, indent=1))
