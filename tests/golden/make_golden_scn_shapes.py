"""Generates tests/golden/scn_shapes.json from the REFERENCE's own source: the constructor arguments and the shape annotations
the author wrote next to the four strided convs of SpMiddleFHD (det3d/models/backbones/scn.py:113,122,134,146, e.g.
`SparseConv3d(16, 32, 3, 2, padding=1, bias=False),  # [41, 1600, 1408] -> [21, 800, 704]`). They are the only
reference-held statement about spconv's output-size rule; tests/test_oracle_sparse_conv_cpu.py holds
oracle.sparse_conv.out_spatial (and the engine's level table) to them. Nothing but these numbers is stored.
Run in the build container only:  python tests/golden/make_golden_scn_shapes.py"""
import ast
import json
import os
import re

REF = "/root/reference/det3d/models/backbones/scn.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rows = []
    pat = re.compile(r"^\s*SparseConv3d\((.*)\),\s*#\s*(\[[^\]]*\])\s*->\s*(\[[^\]]*\])\s*$")
    for lineno, line in enumerate(open(REF), 1):
        if not (100 <= lineno <= 150):   # the SpMiddleFHD constructor
            continue
        m = pat.match(line)
        if not m:
            continue
        call = ast.parse("f(" + m.group(1) + ")").body[0].value
        pos = [ast.literal_eval(a) for a in call.args]
        kw = {k.arg: ast.literal_eval(k.value) for k in call.keywords}
        cin, cout, ksize = pos[0], pos[1], pos[2]
        stride = pos[3] if len(pos) > 3 else kw.get("stride", 1)
        padding = kw.get("padding", 0)
        rows.append(dict(line=lineno, cin=cin, cout=cout, ksize=ksize, stride=stride, padding=padding,
                         in_shape=json.loads(m.group(2)), out_shape=json.loads(m.group(3))))
    assert len(rows) == 4, rows
    json.dump(dict(source="det3d/models/backbones/scn.py", rows=rows), open(os.path.join(HERE, "scn_shapes.json"), "w"), indent=1)
    print(rows)


if __name__ == "__main__":
    main()
