"""One-screen summary of rocprofv3 --pmc passes of the sparse stage (review item: "a one-screen summary instead of 1198 raw lines").

    python scripts/pmc_compact.py LABEL PASS1.csv [PASS2.csv ...] [--trace KERNEL_TRACE.csv] [--tail N] [--match FRAGMENT ...]

Each csv is a *counter_collection.csv of one pass (separate passes for the SQ set, FETCH_SIZE, WRITE_SIZE: they do not fit one
pass). Only the LAST `--tail` dispatches of a pass are used (the eager frames after the autotune sweep), grouped by kernel
(template arguments kept, argument list dropped) and averaged PER LAUNCH. Derived columns:
  wait_any   SQ_WAIT_ANY / SQ_WAVE_CYCLES          share of the wave cycles spent waiting for anything
  wait_inst  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     ... waiting for an instruction to issue (dependencies / issue port)
  mfma_busy  SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CYCLES)   matrix-pipe busy share of the SIMD cycles (4 SIMDs per CU)
  mfma       SQ_INSTS_MFMA per launch
  fetch MB   FETCH_SIZE (KB) * 2 / 1024: the guide's gfx950 correction for 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section)
  write MB   WRITE_SIZE (KB) / 1024 (uncalibrated, as the guide says)"""
import collections
import csv
import re
import sys

args = sys.argv[1:]
label = args.pop(0)
tail, trace, match, files = 400, None, [], []
while args:
    a = args.pop(0)
    if a == "--tail":
        tail = int(args.pop(0))
    elif a == "--trace":
        trace = args.pop(0)
    elif a == "--match":
        match.append(args.pop(0))
    else:
        files.append(a)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    i = name.find("(")
    return (name[:i] if i > 0 else name)[:64]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    keep = set(ids[-tail:])
    for r in rows:
        if int(r["Dispatch_Id"]) not in keep:
            continue
        k = short(r["Kernel_Name"])
        if match and not any(m in k for m in match):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
if trace:
    rows = list(csv.DictReader(open(trace)))
    for r in rows[-tail:]:
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# %s -- per-launch averages over the last %d dispatches of each pass" % (label, tail))
print("%-64s %5s %8s %8s %8s %9s %9s %9s %9s" % ("kernel", "n", "us(prof)", "wait_any", "wait_ins", "mfma_busy", "mfma", "fetch MB", "write MB"))
A = lambda d, c: (sum(d[c]) / len(d[c])) if d.get(c) else None
F = lambda v, fmt: (fmt % v) if v is not None else "-"
for k in sorted(acc, key=lambda k: -(A(acc[k], "SQ_WAVE_CYCLES") or 0) * len(acc[k].get("SQ_WAVE_CYCLES", [1]))):
    d = acc[k]
    wc, busy = A(d, "SQ_WAVE_CYCLES"), A(d, "SQ_BUSY_CYCLES")
    wa, wi, mb = A(d, "SQ_WAIT_ANY"), A(d, "SQ_WAIT_INST_ANY"), A(d, "SQ_VALU_MFMA_BUSY_CYCLES")
    n = len(next(iter(d.values())))
    fe, wr = A(d, "FETCH_SIZE"), A(d, "WRITE_SIZE")
    print("%-64s %5d %8s %8s %8s %9s %9s %9s %9s" % (
        k, n, F(sum(dur[k]) / len(dur[k]) if dur.get(k) else None, "%.1f"), F(wa / wc if wa is not None and wc else None, "%.2f"),
        F(wi / wc if wi is not None and wc else None, "%.2f"), F(mb / (4 * busy) if mb is not None and busy else None, "%.2f"),
        F(A(d, "SQ_INSTS_MFMA"), "%.0f"), F(fe * 2 / 1024 if fe is not None else None, "%.2f"), F(wr / 1024 if wr is not None else None, "%.2f")))
