"""Instruction mix of every loop (backward branch) of one kernel in a hipcc -S listing: python isa_loop_mix.py file.s <symbol substring>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l.split(":")[0]]
for s in start:
    end = next(i for i in range(s, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[s:end]
    print(lines[s].split(":")[0], len(body), "lines")
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if not m:
            continue
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            c = Counter()
            for x in body[labels[t]:i + 1]:
                x = x.strip()
                if not x or x.startswith((".", ";")) or x.endswith(":"):
                    continue
                c[x.split()[0]] += 1
            g = lambda f: sum(v for k, v in c.items() if f(k))
            print(" loop %s lines %d-%d: total %d mfma %d valu %d ds %d buffer %d salu %d" % (
                t, labels[t], i, sum(c.values()), g(lambda k: "mfma" in k), g(lambda k: k.startswith("v_") and "mfma" not in k),
                g(lambda k: k.startswith("ds_")), g(lambda k: k.startswith("buffer_")), g(lambda k: k.startswith("s_"))))
            print("   ", sorted(((v, k) for k, v in c.items() if k.startswith("v_") and "mfma" not in k), reverse=True)[:20])
            print("   ", sorted(((v, k) for k, v in c.items() if k.startswith("s_")), reverse=True)[:10])
