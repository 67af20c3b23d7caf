"""Static check of `hipcc -S` listings for the epilogue pattern that cost the direct conv kernels 10-20 %:
a memory load, `s_waitcnt vmcnt(0)`, a store -- repeated per element, each wait also covering the previous element's store.
Usage: python scripts/isa_serial_epilogue.py file.s [file.s ...]   (hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S x.hip)
Prints every kernel with at least `MIN` (default 6) such load / wait / store groups."""
import re
import sys

MIN = 6


def groups(body):
    lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";")]
    n = 0
    for i, ln in enumerate(lines):
        if ln.startswith("s_waitcnt vmcnt(0)"):
            pre, post = lines[max(0, i - 12):i], lines[i + 1:i + 13]
            if any("global_load" in x or "buffer_load" in x for x in pre) and any("global_store" in x or "buffer_store" in x for x in post):
                n += 1
    return n


for path in sys.argv[1:]:
    text = open(path).read()
    for m in re.finditer(r"^(_Z\S+):\s*; @.*?\n(.*?)s_endpgm", text, re.S | re.M):
        n = groups(m.group(2))
        if n >= MIN:
            print("%-28s %-110s %d" % (path.split("/")[-1], m.group(1)[:110], n))
