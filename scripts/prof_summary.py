"""Summarise a rocprofv3 --kernel-trace rocpd database (*_results.db) into a per-kernel table.

    python scripts/prof_summary.py DB [FRAMES] [ROWS] [MARKER]

FRAMES > 0: only the kernels of the LAST `FRAMES` frames are counted (a frame starts at each launch of the marker kernel: the
frame's clear launch by default; `sparse_pack_batch_kernel` for replays of the captured training iteration), i.e. the timed region
without model set-up, BatchNorm calibration and autotuning; per-frame figures. The last line gives the share of kernels that are
not ours (at::native, rocclr, MIOpen, rocBLAS)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(db.execute("select name, start, end from kernels order by start"))
div = 1.0
if frames > 0:
    if len(sys.argv) > 4:
        marker = sys.argv[4]
    else:
        marker = "fill_multi_kernel" if any("fill_multi_kernel" in r[0] for r in rows) else "vox_insert_kernel"  # a frame's first launch
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    rows = rows[starts[-frames]:]
    div = float(frames)
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for n, s, e in rows:
    d = (e - s) / 1e3
    a = agg[n]
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print("# rocprofv3 --kernel-trace summary of %s" % sys.argv[1].split("/")[-1])
if frames > 0:
    print("# last %d frames: kernel time %.1f us/frame, wall %.1f us/frame" % (frames, tot / div, (rows[-1][2] - rows[0][1]) / 1e3 / div))
else:
    print("# total kernel time %.1f us" % tot)
print("%-100s %9s %11s %9s %9s %9s %6s" % ("kernel", "calls/fr" if frames else "calls", "us/frame" if frames else "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, a in sorted(agg.items(), key=lambda x: -x[1][1])[:nrows]:
    print("%-100s %9.1f %11.1f %9.1f %9.1f %9.1f %5.1f%%" % (n[:100], a[0] / div, a[1] / div, a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
import re
foreign = sum(a[1] for n, a in agg.items() if re.search(r"at::native|rocclr|miopen|igemm|naive_conv|batched_transpose|SubTensor|Cijk_|rocprim|hipcub", n))
print("# kernels not from libsessd_hip.so (at::native, rocclr, MIOpen, rocBLAS): %.1f us%s = %.2f %% of the kernel time; %d launches%s" % (
    foreign / div, "/frame" if frames else "", 100 * foreign / tot, sum(a[0] for a in agg.values()) / div, "/frame" if frames else ""))

# the launches around the longest instance of a kernel (argument 5: a name fragment), to see WHICH call it is
if len(sys.argv) > 5:
    frag = sys.argv[5]
    hits = [(e - s_, i) for i, (n, s_, e) in enumerate(rows) if frag in n]
    if hits:
        _, at = max(hits)
        print("# around the longest %s launch:" % frag)
        for i in range(max(0, at - 4), min(len(rows), at + 4)):
            n, s_, e = rows[i]
            print("#   %s %8.1f us  %s" % ("->" if i == at else "  ", (e - s_) / 1e3, n[:110]))
