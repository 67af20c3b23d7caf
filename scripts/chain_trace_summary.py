"""Per-launch durations (us) of the chain_* / fill kernels from a rocprofv3 --kernel-trace csv, in launch order per kernel."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(r"chain_\w+|fill_u32", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = {}
for r in rows:
    k = re.search(r"(chain_\w+|fill_u32)", r["Kernel_Name"]).group(1)
    out.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in out.items():
    print("%-24s" % k, " ".join("%6.1f" % x for x in v))
