"""Concurrency of a multi-stream run from a rocprofv3 --kernel-trace CSV: python timeline_overlap.py kernel_trace.csv [last_frames]
Splits the wall time of the last `last_frames` frames (a frame starts at a vox_insert_kernel launch) into: idle, one kernel alone
(by class), two or more kernels concurrently."""
import csv
import sys
from collections import defaultdict

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0"))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 100
starts = [r[0] for r in rows if "vox_insert_kernel" in r[2]]
t_lo = starts[-nlast - 2]
t_hi = starts[-2]          # drop the tail (end-of-job gather, the last frames draining)
rows = [r for r in rows if t_lo <= r[0] < t_hi]


def cls(n):
    if "winograd" in n or "conv2d_mfma" in n or "ssfa_fuse" in n:
        return "dense"
    if "sparse_conv" in n:
        return "sparse_conv"
    if "chain_" in n or "vox_" in n or "fill_" in n or "stage_points" in n:
        return "sites/voxelize"
    return "predict/other"


ev = []
for s, e, n, q, st in rows:
    ev.append((s, 1, cls(n)))
    ev.append((e, -1, cls(n)))
ev.sort()
active = defaultdict(int)
acc = defaultdict(float)
last = ev[0][0]
for t, d, c in ev:
    dt = t - last
    if dt > 0:
        live = sorted(k for k, v in active.items() if v > 0)
        n = sum(active.values())
        key = "idle" if n == 0 else ("alone: " + live[0] if n == 1 else "concurrent: " + "+".join(live))
        acc[key] += dt
    active[c] += d
    last = t
tot = sum(acc.values())
frames = sum(1 for r in rows if "vox_insert_kernel" in r[2])
print("# %d kernels, %d frames, wall %.1f us/frame; queues %s" % (len(rows), frames, tot / 1e3 / max(frames, 1), sorted(set(r[3] for r in rows))))
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print("%-50s %8.1f us/frame %5.1f%%" % (k, v / 1e3 / max(frames, 1), 100 * v / tot))
