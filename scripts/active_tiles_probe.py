"""Times the first three SSFA layers (3x3 128->128 @200x176) dense (tile_cfg 22 / 23) against the active-tile mode
(sessd_bev_tile_activity + sessd_fill_inactive_tiles + sessd_conv3x3_winograd_sk_active) on the BEV occupancy of synthetic
20 k-point scans (the site pixels come from a CPU restatement of the strided site rule; random features and weights -- only the
geometry matters for the time). Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
import numpy as np
import torch

from oracle import capi
from sessd_hip import ops, synth

H, W, C = 200, 176, 128
dev = torch.device("cuda:0")


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


def l4_sites(seed, batch, npts=20000, ss=1, mv=16000):
    rows = []
    for b in range(batch):
        pts = synth.make_frame(seed + b, npts, supersample=ss)
        _, c, _ = capi.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, mv)
        shape = [41, 1600, 1408]
        co = c
        for (k, s, p) in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                          ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
            co, shape = down(co, shape, k, s, p)
        rows.append(np.concatenate([np.full((len(co), 1), b), co], 1))
    return np.ascontiguousarray(np.concatenate(rows).astype(np.int32))


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = {}
for tag, batch, kw in (("batch1_20k", 1, {}), ("batch8_200k", 8, dict(npts=None, ss=3, mv=64000))):
    idx = l4_sites(1, batch, **kw)
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(batch, C, H, W)
    x[idx[:, 0], :, idx[:, 2], idx[:, 3]] = torch.randn(len(idx), C, generator=g)
    x = x.to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
    sc, sh = (0.5 + torch.rand(C, generator=g)).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    pc = ops.pack_conv2d(w)
    ta = ops.TileActivity(batch, H, W, 3, dev)
    di, n = torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev)
    ta.run(di, n, len(idx))
    tiles = batch * (H // 2) * (W // 2)
    res = {"l4_sites": int(len(idx)), "active_tile_fraction": [float(v) / tiles for v in ta.n_list.cpu()]}
    outs = [torch.zeros(batch, C, H, W, device=dev) for _ in range(3)]
    consts = [torch.rand(C, device=dev) for _ in range(3)]
    res["activity_us"] = timeit(lambda: ta.run(di, n, len(idx)))
    res["fill_3_layers_us"] = timeit(lambda: ta.fill(outs, consts))
    for shape in (0, 1):
        ws = torch.zeros(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(batch, H, W, C, shape, 0)), dtype=torch.uint8, device=dev)
        dense_out = torch.zeros(batch, C, H, W, device=dev)
        r = {"dense_us": timeit(lambda: ops.conv2d(x, pc, sc, sh, True, None, dense_out, 22 + shape, workspace=ws))}
        for l in range(3):
            for mr in (1, 2, 4, 8):
                r["layer%d_min_rounds_%d_us" % (l, mr)] = timeit(lambda: ops.conv2d_winograd_sk_active(
                    x, pc.upk_sk(shape), C, sc, sh, True, outs[l], shape, ws, ta.tile_list[l], ta.n_list[l:l + 1], min_rounds=mr))
        res["tile_cfg_%d" % (22 + shape)] = r
    out[tag] = res
print(json.dumps(out))
