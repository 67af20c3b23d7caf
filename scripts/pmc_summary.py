"""Average PMC counters per kernel from a rocprofv3 --pmc csv (counter_collection.csv)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s n=%3d avg=%.4g" % (c, len(v), sum(v) / len(v)))
