"""CPU only: the share of tiles every SSFA layer would compute in active-tile mode on synthetic scans -- the rule of
csrc/dense_active.hip (tests/test_active_rule_cpu.py::masks) applied to the BEV occupancy of the last sparse level (C oracle
voxelizer + the strided site rule), continued PAST the layers the engine runs over lists: what a list launch of conv_0 / conv_1
(behind the transposed convs) would have to compute. Prints one JSON line; `profiles/r4s2_active_tile_fractions_cpu.json`.

    python scripts/active_tile_stats.py [--stress]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "se-ssd_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

from oracle import capi
from sessd_hip import synth
from test_active_rule_cpu import masks

H, W = 200, 176


def down(coords, shape, k, s, p):
    od = [(shape[i] + 2 * p[i] - k[i]) // s[i] + 1 for i in range(3)]
    out = np.zeros(od, bool)
    zs, ys, xs = coords[:, 0], coords[:, 1], coords[:, 2]
    for kz in range(k[0]):
        for ky in range(k[1]):
            for kx in range(k[2]):
                oz, oy, ox = zs + p[0] - kz, ys + p[1] - ky, xs + p[2] - kx
                m = (oz % s[0] == 0) & (oy % s[1] == 0) & (ox % s[2] == 0)
                oz, oy, ox = oz[m] // s[0], oy[m] // s[1], ox[m] // s[2]
                m = (oz >= 0) & (oz < od[0]) & (oy >= 0) & (oy < od[1]) & (ox >= 0) & (ox < od[2])
                out[oz[m], oy[m], ox[m]] = True
    return np.argwhere(out), od


def bev_sites(seed, npts, ss, mv):
    pts = synth.make_frame(seed, npts, supersample=ss)
    _, co, _ = capi.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, mv)
    shape = [41, 1600, 1408]
    for (k, s, p) in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                      ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
        co, shape = down(co, shape, k, s, p)
    nc = np.zeros((H, W), bool)
    nc[co[:, 1], co[:, 2]] = True
    return nc


def conv_tiles(nc):
    """tiles of a 3x3 stride-1 layer over a map whose non-constant pixels are nc (non-zero constant: border ring)"""
    h, w = nc.shape
    p = np.pad(nc, 1)
    tm = np.zeros((h // 2, w // 2), bool)
    for dy in range(4):
        for dx in range(4):
            tm |= p[dy:dy + h:2, dx:dx + w:2][:h // 2, :w // 2]
    tm[0, :] = tm[-1, :] = True
    tm[:, 0] = tm[:, -1] = True
    return tm


stress = "--stress" in sys.argv
kw = dict(npts=None, ss=3, mv=64000) if stress else dict(npts=20000, ss=1, mv=16000)
names = ["b0.0", "b0.1", "b0.2", "b1.0", "b1.1", "b1.2", "deconv_0+deconv_1"]
rows = []
for seed in range(8):
    nc0 = bev_sites(seed, **kw)
    m = masks(nc0, [0, 0, 0, 2, 0, 0, 3])
    r = {"site_pixels": float(nc0.mean())}
    r.update({n: float(t.mean()) for n, t in zip(names, m)})
    r["trans_0"], r["trans_1"] = r["b0.2"], r["b1.2"]
    # the maps behind the transposed convs are not constant in the 4x4 blocks of the pair's tiles
    mid = m[6].repeat(4, 0).repeat(4, 1)
    r["conv_0 (if listed)"] = r["conv_1 (if listed)"] = float(conv_tiles(mid).mean())
    rows.append(r)
out = {"workload": "dense-scene frames (200 k points, <= 64 k voxels)" if stress else "20 k-point frames (<= 16 k voxels)", "frames": len(rows),
       "mean": {k: round(float(np.mean([r[k] for r in rows])), 4) for k in rows[0]},
       "min": {k: round(float(np.min([r[k] for r in rows])), 4) for k in rows[0]},
       "max": {k: round(float(np.max([r[k] for r in rows])), 4) for k in rows[0]}}
print(json.dumps(out))
