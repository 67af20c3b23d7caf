"""Kernel sequence of the LAST frame of a single-stream rocprofv3 --kernel-trace CSV (start, duration, gap to the previous kernel)."""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "vox_insert_kernel" in r[2]]
a, b = starts[-3], starts[-2]
a = max(a - 6, 0)
prev = rows[a][0]
for s, e, n in rows[a:b]:
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - rows[a][0]) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:90]))
    prev = e
